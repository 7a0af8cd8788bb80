"""Drop-in for the reference's ``nlp_classifier`` module (nlp_classifier.py:6-42): text tower + ArcFace head.

Same constructor / attributes / ``forward`` / ``predict_emb`` as the reference; ``forward_loss`` is the fused
training path (margin + scaled cross-entropy + argmax without materialising the logits).
"""
import torch.nn as nn

from arcface import ArcMarginProduct
from transformer_emb import TransformerEmb
from multimodalsimilar_amd.bert import as_native


class NlpClassifier(nn.Module):
    def __init__(self, pretrained_model, num_labels, emb_size=128, dropout=None):
        super().__init__()
        self.ptm = as_native(pretrained_model)
        self.dropout = nn.Dropout(p=dropout if dropout is not None else 0.1)   # never applied (reference :10)
        self.num_labels = num_labels
        self.emb_size = emb_size
        self.emb_layer = TransformerEmb(self.ptm, self.emb_size)
        self.classifier = ArcMarginProduct(self.ptm.config.hidden_size, self.num_labels)   # s=64, m=0.40

    def forward(self, query_input_ids, query_token_type_ids=None, query_position_ids=None, query_attention_mask=None,
                label=None, is_test=False):
        emb = self.emb_layer(query_input_ids, query_token_type_ids, query_position_ids, query_attention_mask)
        if not is_test:
            return self.classifier(emb, label)
        return self.classifier.forward_test(emb)

    def forward_loss(self, query_input_ids, query_token_type_ids=None, query_position_ids=None,
                     query_attention_mask=None, label=None):
        emb = self.emb_layer(query_input_ids, query_token_type_ids, query_position_ids, query_attention_mask)
        return self.classifier.forward_loss(emb, label)

    def predict_emb(self, query_input_ids, query_token_type_ids=None, query_position_ids=None, query_attention_mask=None):
        return self.emb_layer(query_input_ids, query_token_type_ids, query_position_ids, query_attention_mask)
