"""CPU oracle for the two-tower + ArcFace training step.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch fp32 CPU restatement of the
reference's algorithm (forrestsocool/MultimodalSimilar) for the hot path named in
BASELINE.json.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package ``multimodalsimilar_amd`` never
does, and fails loudly when its HIP library is missing instead of falling back to this.

Pinning status (see DESIGN.md "Oracle"):
  * arcface_ref      - pinned: checked against outputs of the reference's own ``arcface.py``
                       run in the build container (tests/golden/arcface_*.npz).
  * bert_ref         - pinned: checked against the reference's ``NlpClassifier`` over HF
                       ``BertModel`` (tests/golden/nlp_*.npz).
  * glue (normalise+concat) - pinned through the arcface goldens composed per
                       multimodal_classifier.py:54-56 (tests/golden/glue_*.npz).
  * effnet_ref       - PARITY UNPINNED: the reference delegates the image tower to ``timm``,
                       which is neither vendored in the reference nor installed here, and the
                       reference's tests hold no vectors for it.  effnet_ref restates timm's
                       published ``efficientnet_b0/b4`` architecture.  Its block
                       definitions (op order, squeeze-excite placement and width, channel rounding, repeats,
                       skip rule) are cross-checked to float rounding against an independent implementation that
                       IS installed, ``transformers.models.efficientnet``, with the oracle in its
                       ``variant(tf_same=True, bn_eps=1e-3)`` mode (tests/test_oracle_effnet_hf.py): that is
                       the TF-port variant, not timm's numerics, so the status stays "parity unpinned".
  * search_ref       - PARITY UNPINNED: the similarity search after the model (nlp_infer.py:139-152) is ``faiss``
                       (no pinned version, not installed, no vectors in the reference's tests); restates
                       IndexFlat / METRIC_INNER_PRODUCT semantics, equal scores by ascending index.
  * multilabel       - pinned: bert_ref + three arcface_ref heads against the reference's
                       ``NlpClassifierMultilabel`` run here (tests/golden/nlp_multilabel.npz).
  * optim_ref        - pinned against torch.optim.AdamW + transformers.get_scheduler("linear"),
                       the exact objects the reference's train script constructs.
"""
