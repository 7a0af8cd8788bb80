"""Drop-in for the reference's ``arcface`` module: ``arcface.ArcMarginProduct`` on the MI355X HIP path.

Same constructor, attributes and methods as /root/reference/arcface.py:17-67 (whole-module pickles resolve
``arcface.ArcMarginProduct``); the arithmetic runs in libmmsim_hip.so (multimodalsimilar_amd/head.py).
"""
from multimodalsimilar_amd.head import ArcMarginProduct  # noqa: F401

__all__ = ["ArcMarginProduct"]
