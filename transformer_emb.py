"""Drop-in for the reference's ``transformer_emb`` module (transformer_emb.py:6-30).

``TransformerEmb(pretrained_model, emb_size)`` returns the text tower's ``pooler_output``.  The reference keeps
two layers it never applies (``emb_layer`` Linear, ``bn_layer`` BatchNorm1d, transformer_emb.py:12-13); they are
kept here too so state dicts and pickles interchange, and like there they never receive gradients.
``pretrained_model`` may be a ``multimodalsimilar_amd.bert.BertModel`` or any HF-style BERT module exposing
``.config`` and HF parameter names: the latter is converted once to the native tower (weights copied).
"""
import torch.nn as nn

from multimodalsimilar_amd.bert import BertModel, BertConfig, as_native  # noqa: F401


class TransformerEmb(nn.Module):
    def __init__(self, pretrained_model, emb_size=128, dropout=None):
        super().__init__()
        self.ptm = as_native(pretrained_model)
        self.dropout = nn.Dropout(p=dropout if dropout is not None else 0.1)
        self.emb_layer = nn.Linear(self.ptm.config.hidden_size, emb_size)      # unused, as in the reference
        self.bn_layer = nn.BatchNorm1d(self.ptm.config.hidden_size)            # unused, as in the reference

    def forward(self, query_input_ids, query_token_type_ids=None, query_position_ids=None, query_attention_mask=None):
        outputs = self.ptm(input_ids=query_input_ids, attention_mask=query_attention_mask,
                           token_type_ids=query_token_type_ids, position_ids=query_position_ids)
        return outputs.pooler_output
