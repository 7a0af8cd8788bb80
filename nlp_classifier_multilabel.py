"""Drop-in for the reference's ``nlp_classifier_multilabel`` module (nlp_classifier_multilabel.py:6-53):
one text-tower embedding shared by three ArcFace heads (first category m=0.4, second category m=0.2, tag m=0.1).

Same constructor / attributes (ptm, dropout, emb_size, emb_layer, firstcate_classifier, secondcate_classifier,
tag_classifier) / ``forward`` (a 3-tuple of logits, margin logits in training and plain cosines with ``is_test``) /
``predict_emb``.  ``forward_loss`` is the fused training path: the weighted sum of the three margin cross-entropies the
reference's script forms (nlp_classifier_train_daodian_v3_dist.py:164-166) without materialising any [B, C] logits,
plus the three argmax predictions.
"""
import torch.nn as nn

from arcface import ArcMarginProduct
from transformer_emb import TransformerEmb
from multimodalsimilar_amd.bert import as_native


class NlpClassifierMultilabel(nn.Module):
    MARGINS = (0.4, 0.2, 0.1)       # first category, second category, tag (reference :15-17)

    def __init__(self, pretrained_model, firstcate_num_labels, secondcate_num_labels, tag_num_labels, emb_size=128, dropout=None):
        super().__init__()
        tower = as_native(pretrained_model)
        self.ptm = tower
        self.dropout = nn.Dropout(p=0.1 if dropout is None else dropout)         # created, never applied (reference :10)
        self.emb_size = emb_size
        self.emb_layer = TransformerEmb(tower)                                   # the reference passes no emb_size here (:14)
        width = tower.config.hidden_size
        sizes = (firstcate_num_labels, secondcate_num_labels, tag_num_labels)
        first, second, tag = (ArcMarginProduct(width, n, m=m) for n, m in zip(sizes, self.MARGINS))
        self.firstcate_classifier, self.secondcate_classifier, self.tag_classifier = first, second, tag

    def _heads(self):
        return self.firstcate_classifier, self.secondcate_classifier, self.tag_classifier

    def forward(self, query_input_ids, query_token_type_ids=None, query_position_ids=None, query_attention_mask=None,
                firstcate_label=None, secondcate_label=None, tag_label=None, is_test=False):
        emb = self.emb_layer(query_input_ids, query_token_type_ids, query_position_ids, query_attention_mask)
        if not is_test:
            return tuple(h(emb, y) for h, y in zip(self._heads(), (firstcate_label, secondcate_label, tag_label)))
        return tuple(h.forward_test(emb) for h in self._heads())

    def forward_loss(self, query_input_ids, query_token_type_ids=None, query_position_ids=None, query_attention_mask=None,
                     firstcate_label=None, secondcate_label=None, tag_label=None, weights=(1.0, 1.0, 1.0)):
        """-> (w1 CE1 + w2 CE2 + w3 CE3, (argmax1, argmax2, argmax3)); each CE is the mean margin cross-entropy of one head."""
        emb = self.emb_layer(query_input_ids, query_token_type_ids, query_position_ids, query_attention_mask)
        total, preds = None, []
        for h, y, w in zip(self._heads(), (firstcate_label, secondcate_label, tag_label), weights):
            loss, pred = h.forward_loss(emb, y)
            total = loss * w if total is None else total + loss * w
            preds.append(pred)
        return total, tuple(preds)

    def predict_emb(self, query_input_ids, query_token_type_ids=None, query_position_ids=None, query_attention_mask=None):
        return self.emb_layer(query_input_ids, query_token_type_ids, query_position_ids, query_attention_mask)
