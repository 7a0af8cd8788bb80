"""Drop-in for the reference's ``nlp_classifier`` module (nlp_classifier.py:6-42): text tower + ArcFace head.

Constructor, attributes (``ptm``, ``dropout``, ``num_labels``, ``emb_size``, ``emb_layer``, ``classifier``), ``forward`` and
``predict_emb`` keep the reference's names and argument meaning; everything below them is the MI355X path:

* ``ptm`` is converted once into the native text tower (``multimodalsimilar_amd.bert.BertModel``: flat fp32 / bf16 parameter
  buffers, hand-written HIP kernels behind ``libmmsim_hip.so``); an HF ``BertModel`` is accepted and its weights are copied.
* ``classifier`` is the HIP ArcFace head (``arcface.ArcMarginProduct``: s = 64, m = 0.40 as in the reference's default).
* ``forward_loss`` is an additive, non-breaking extension: margin + scaled cross-entropy + argmax in one fused pass that
  never materialises the [B, C] logits (what the shipped train script uses).
* There is no CPU fallback: inputs off the GPU raise ``MmsimError``.
"""
import torch.nn as nn

from arcface import ArcMarginProduct
from multimodalsimilar_amd.bert import as_native
from transformer_emb import TransformerEmb


class NlpClassifier(nn.Module):
    def __init__(self, pretrained_model, num_labels, emb_size=128, dropout=None):
        super().__init__()
        tower = as_native(pretrained_model)
        p_drop = 0.1 if dropout is None else dropout
        self.ptm = tower
        self.dropout = nn.Dropout(p=p_drop)                  # created and never applied, as in the reference (:10)
        self.num_labels, self.emb_size = num_labels, emb_size
        self.emb_layer = TransformerEmb(tower, emb_size)     # registers the tower a second time (state-dict keys)
        self.classifier = ArcMarginProduct(tower.config.hidden_size, num_labels)

    def _embed(self, ids, token_types, positions, mask):
        return self.emb_layer(ids, token_types, positions, mask)

    def forward(self, query_input_ids, query_token_type_ids=None, query_position_ids=None, query_attention_mask=None,
                label=None, is_test=False):
        """Margin logits [B, C] for training (``label`` required), plain cosines with ``is_test`` (nlp_classifier.py:17-31)."""
        e = self._embed(query_input_ids, query_token_type_ids, query_position_ids, query_attention_mask)
        return self.classifier.forward_test(e) if is_test else self.classifier(e, label)

    def forward_loss(self, query_input_ids, query_token_type_ids=None, query_position_ids=None,
                     query_attention_mask=None, label=None):
        """-> (mean cross-entropy of the margin logits, argmax) on the fused head path."""
        e = self._embed(query_input_ids, query_token_type_ids, query_position_ids, query_attention_mask)
        return self.classifier.forward_loss(e, label)

    def predict_emb(self, query_input_ids, query_token_type_ids=None, query_position_ids=None, query_attention_mask=None):
        """Pooled text embedding [B, hidden] in (-1, 1) (nlp_classifier.py:33-42)."""
        return self._embed(query_input_ids, query_token_type_ids, query_position_ids, query_attention_mask)
