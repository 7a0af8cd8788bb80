"""Drop-in for the reference's ``multimodal_classifier`` module (multimodal_classifier.py:13-57): the two-tower
model -- image tower + text tower, each embedding L2-normalised, concatenated, ArcFace(m=0.5) head.

Same constructor / attributes (cv, nlp, classifier, dropout, num_labels, emb_size) / ``forward`` / ``predict_emb``.
``cv_classifier_path`` / ``nlp_classifier_path`` may be whole-module pickle paths as in the reference
(``torch.load(..., weights_only=False)``, E10) or already-constructed CvClassifier / NlpClassifier modules
(offline there is nothing to download or unpickle).  ``forward_loss`` is the fused training path.
"""
import torch
import torch.nn as nn

from arcface import ArcMarginProduct
from multimodalsimilar_amd.head import glue_concat


import os
_TWO_STREAMS = os.environ.get("MMSIM_TWO_STREAMS", "1") != "0"      # 0: both towers on the caller's stream
_BWD_TEXT_FIRST = os.environ.get("MMSIM_BWD_ORDER", "text") != "image"      # which tower's backward autograd replays first


def _load_tower(obj):
    if isinstance(obj, nn.Module):
        return obj
    return torch.load(obj, weights_only=False)          # multimodal_classifier.py:16-17


class MultimodalClassifier(nn.Module):
    def __init__(self, device, cv_classifier_path, nlp_classifier_path, emb_size, num_labels, dropout=None):
        super().__init__()
        self.cv = _load_tower(cv_classifier_path)
        self.nlp = _load_tower(nlp_classifier_path)
        self.dropout = nn.Dropout(p=dropout if dropout is not None else 0.1)     # never applied (reference :18, E6)
        self.num_labels = num_labels
        self.emb_size = emb_size
        self.classifier = ArcMarginProduct(in_feature=self.emb_size, out_feature=self.num_labels, m=0.5)   # :22
        self.cv.to(device)
        self.nlp.to(device)
        self.classifier.to(device)

    def forward(self, img_input: torch.Tensor, query_input_ids, query_token_type_ids=None, query_position_ids=None,
                query_attention_mask=None, label=None, is_test=False):
        final_embdding = self.predict_emb(img_input=img_input, query_input_ids=query_input_ids,
                                          query_token_type_ids=query_token_type_ids,
                                          query_attention_mask=query_attention_mask)
        if not is_test:
            return self.classifier(final_embdding, label)
        return self.classifier.forward_test(final_embdding)

    def forward_loss(self, img_input, query_input_ids, query_token_type_ids=None, query_position_ids=None,
                     query_attention_mask=None, label=None):
        """(mean cross-entropy of the margin logits, argmax) without materialising the [B, C] logits."""
        emb = self.predict_emb(img_input, query_input_ids, query_token_type_ids, None, query_attention_mask)
        return self.classifier.forward_loss(emb, label)

    def predict_emb(self, img_input: torch.Tensor, query_input_ids, query_token_type_ids=None, query_position_ids=None,
                    query_attention_mask=None):
        # The towers are independent until the glue: the image tower runs on a second HIP stream so that its many small,
        # latency-bound kernels (7x7 / 14x14 stages, reductions) fill in under the text tower's GEMMs.  autograd replays
        # each tower's backward on the stream its forward ran on and orders the hand-offs, so backward overlaps as well.
        if img_input.is_cuda and _TWO_STREAMS:
            main = torch.cuda.current_stream(img_input.device)
            side = self._side_stream(img_input.device)
            side.wait_stream(main)
            # Host launch order matters as much as the streams: the text tower is ~250 launches of long kernels, the image
            # tower ~550 launches of short ones (~14 ms of host time), and whichever is enqueued second starts that late.
            # Forward: text first.  Backward: autograd runs the node created LAST first, so the image node must be created
            # first -- its launches are deferred (node now, kernels after the text tower's).
            backbone = getattr(self.cv, "backbone", None)
            defer = (torch.is_grad_enabled() and hasattr(backbone, "defer_launches") and not getattr(self.cv, "use_fc", True)
                     and _BWD_TEXT_FIRST)
            if not _BWD_TEXT_FIRST and torch.is_grad_enabled():
                # image backward replayed first: plain order text -> image in forward (the image node is the younger one)
                title_embedding = self.nlp.predict_emb(query_input_ids=query_input_ids, query_token_type_ids=query_token_type_ids,
                                                       query_attention_mask=query_attention_mask)
                with torch.cuda.stream(side):
                    img_embedding = self.cv.predict_emb(img_input)
            elif defer or not torch.is_grad_enabled():
                # the whole defer -> text tower -> flush sequence is one unit: if anything in it raises (CPU ids, S beyond the
                # position table, MmsimError), the tower must not stay armed -- a later standalone forward would otherwise
                # return an unwritten tensor
                try:
                    if defer:
                        backbone.defer_launches()
                        with torch.cuda.stream(side):
                            img_embedding = self.cv.predict_emb(img_input)          # autograd node only
                    title_embedding = self.nlp.predict_emb(query_input_ids=query_input_ids, query_token_type_ids=query_token_type_ids,
                                                           query_attention_mask=query_attention_mask)
                    with torch.cuda.stream(side):
                        if defer:
                            backbone.flush_deferred()
                        else:
                            img_embedding = self.cv.predict_emb(img_input)
                finally:
                    if defer:
                        backbone._defer = None
            else:
                with torch.cuda.stream(side):
                    img_embedding = self.cv.predict_emb(img_input)
                title_embedding = self.nlp.predict_emb(query_input_ids=query_input_ids, query_token_type_ids=query_token_type_ids,
                                                       query_attention_mask=query_attention_mask)
            main.wait_stream(side)
            img_embedding.record_stream(main)
            img_input.record_stream(side)
        else:
            img_embedding = self.cv.predict_emb(img_input)
            title_embedding = self.nlp.predict_emb(query_input_ids=query_input_ids, query_token_type_ids=query_token_type_ids,
                                                   query_attention_mask=query_attention_mask)   # position ids dropped (E13)
        return glue_concat(img_embedding, title_embedding)                                   # :54-56

    def __getstate__(self):                 # whole-module pickles (torch.save(model)) must not carry the stream handle
        d = self.__dict__.copy()
        d.pop("_side", None)
        return d

    def _side_stream(self, device):
        s = getattr(self, "_side", None)
        if s is None or s.device != torch.device(device):
            s = torch.cuda.Stream(device=device, priority=int(os.environ.get("MMSIM_SIDE_PRIORITY", "0")))
            object.__setattr__(self, "_side", s)          # not a module attribute: never pickled / moved
            from multimodalsimilar_amd import ops
            ops.register_side_stream(s)                   # FusedAdamW.step / GradientExchange.finish wait for it
        return s
