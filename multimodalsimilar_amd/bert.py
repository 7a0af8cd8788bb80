"""MI355X-native BERT text tower (the arithmetic of HF ``BertModel`` that transformer_emb.py:20-24 calls).

``BertModel`` here is a host-side schedule over the HIP kernels of libmmsim_hip.so: it owns one flat parameter
buffer (fp32 master + fp32 grad + bf16 shadow), a per-(B,S) activation workspace in HBM, and runs the whole
encoder forward / backward as explicit kernel sequences.  Parameter names and shapes are HF's
(``embeddings.word_embeddings.weight`` ... ``pooler.dense.bias``) so state dicts interchange.

Sequence per layer (reference lines: transformers modeling_bert.py):
  fwd  QKV GEMM(+bias) -> fused attention -> O GEMM(+bias) -> dropout+residual+LN -> FFN1 GEMM(+bias, GELU)
       -> FFN2 GEMM(+bias) -> dropout+residual+LN                                     (:164-203, 289-293, 334-351)
  bwd  LN bwd (+dgamma, dbeta, dbias) -> wgrad/dgrad GEMMs (GELU' and the residual add fused in the dgrad
       epilogues) -> fused attention bwd -> QKV wgrad/dgrad.
Gradients are written straight into the flat gradient buffer (parameter ``.grad`` are views of it).
"""
import math
from types import SimpleNamespace

import os
import torch
import torch.nn as nn

from . import ops
from .flat import FlatBuffer
from ._lib import MmsimError


# q|k|v bias gradients out of attn_bwd (MMSIM_FUSE_BIAS_GRADS=0: separate column-sum pass).  The same fusion for the
# intermediate.dense bias in the dX GEMM epilogue measured SLOWER on cfg4 (+1 ms/step: the epilogue is on the GEMM's critical
# path, the column-sum pass it saves overlaps with the image tower's stream anyway) and was removed.
_FUSE_ATTN_BIAS = os.environ.get("MMSIM_FUSE_BIAS_GRADS", "1") != "0"
# MMSIM_WGRAD_PAIR=0: the attention-output and q|k|v weight gradients of a layer as two launches instead of one grouped launch
_WGRAD_STREAM = os.environ.get("MMSIM_WGRAD_STREAM", "0") != "0"
_WGRAD_PAIR = os.environ.get("MMSIM_WGRAD_PAIR", "1") != "0"
# MMSIM_GELU_PAIR=0: the intermediate activation keeps its pre-activation and the dgrad epilogue recomputes gelu' from it
# (epilogues 1 / 2) instead of saving gelu' in forward and multiplying in backward (epilogues 6 / 7); A/B switch
_GELU_PAIR = os.environ.get("MMSIM_GELU_PAIR", "1") != "0"

class BertConfig:
    """The subset of HF BertConfig the tower needs (defaults: hfl/chinese-roberta-wwm-ext)."""

    def __init__(self, vocab_size=21128, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, initializer_range=0.02, **_):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.layer_norm_eps = layer_norm_eps
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.initializer_range = initializer_range

    @classmethod
    def roberta_wwm_ext_base(cls, **kw):
        return cls(**kw)

    @classmethod
    def roberta_wwm_ext_large(cls, **kw):
        return cls(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, **kw)


class _Holder(nn.Module):
    """Name-space node so that parameters carry HF's dotted names."""


def _register(root, dotted, param):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Holder())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


def _layer_specs(cfg, i):
    H, I = cfg.hidden_size, cfg.intermediate_size
    p = f"encoder.layer.{i}."
    return [  # q|k|v weights, then q|k|v biases, are adjacent: the fused [3H,H] / [3H] views
        (p + "attention.self.query.weight", (H, H)), (p + "attention.self.key.weight", (H, H)),
        (p + "attention.self.value.weight", (H, H)),
        (p + "attention.self.query.bias", (H,)), (p + "attention.self.key.bias", (H,)),
        (p + "attention.self.value.bias", (H,)),
        (p + "attention.output.dense.weight", (H, H)), (p + "attention.output.dense.bias", (H,)),
        (p + "attention.output.LayerNorm.weight", (H,)), (p + "attention.output.LayerNorm.bias", (H,)),
        (p + "intermediate.dense.weight", (I, H)), (p + "intermediate.dense.bias", (I,)),
        (p + "output.dense.weight", (H, I)), (p + "output.dense.bias", (H,)),
        (p + "output.LayerNorm.weight", (H,)), (p + "output.LayerNorm.bias", (H,)),
    ]


class _Workspace:
    """Activation / gradient scratch in HBM for one (B, S); reused every step (stable addresses)."""

    def __init__(self, cfg, B, S, device, keep_layers):
        H, I, L, nh = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads
        M = B * S
        bf, f32 = torch.bfloat16, torch.float32
        e = lambda *s, dt=bf: torch.empty(*s, dtype=dt, device=device)
        n = L if keep_layers else 1
        self.keep = keep_layers
        self.h = [e(M, H) for _ in range(n + 1)]
        self.qkv = [e(M, 3 * H) for _ in range(n)]
        self.ctx = [e(M, H) for _ in range(n)]
        self.y1 = [e(M, H) for _ in range(n)]
        self.h1 = [e(M, H) for _ in range(n)]
        self.upre = [e(M, I) for _ in range(n)]
        self.u = [e(M, I) for _ in range(n)]
        self.y2 = [e(M, H) for _ in range(n)]
        self.lse = [e(B * nh * S, dt=f32) for _ in range(n)]
        self.st = [e(4, M, dt=f32) for _ in range(n)]     # mean1, rstd1, mean2, rstd2
        self.gen = 0          # bumped by every forward that writes this workspace; an autograd context remembers its value
        self.t = e(M, H)
        self.cls = e(B, H)
        self.pooled = e(B, H, dt=f32)
        if keep_layers:
            self.dh = [e(M, H), e(M, H)]
            self.dy = [e(M, H), e(M, H)]
            self.dt = e(M, H)
            self.dt2 = e(M, H)                       # second dropped-gradient buffer: lets the weight-gradient stream read one while the next is written
            self.dhb = e(M, H)
            self.du = e(M, I)
            self.dqkv = e(M, 3 * H)
            self.dctx = e(M, H)
            self.dpre = e(B, H)
            self.dcls = e(B, H)


class BertModel(nn.Module):
    def __init__(self, config, device=None, seed=None):
        super().__init__()
        self.config = config
        cfg = config
        H = cfg.hidden_size
        if H != cfg.num_attention_heads * 64:
            raise ValueError("the HIP attention kernels need head_dim == 64 (hidden_size == 64 * heads)")
        specs = [("embeddings.word_embeddings.weight", (cfg.vocab_size, H)),
                 ("embeddings.position_embeddings.weight", (cfg.max_position_embeddings, H)),
                 ("embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, H)),
                 ("embeddings.LayerNorm.weight", (H,)), ("embeddings.LayerNorm.bias", (H,))]
        for i in range(cfg.num_hidden_layers):
            specs += _layer_specs(cfg, i)
        specs += [("pooler.dense.weight", (H, H)), ("pooler.dense.bias", (H,))]
        self._flat = FlatBuffer(specs, device="cpu")
        self._init_weights(seed)
        for name, _ in specs:
            _register(self, name, nn.Parameter(self._flat.view(name)))
        # autograd anchor: lets loss.backward() reach the tower although its inputs are integer ids
        self._anchor = torch.zeros((), requires_grad=True)
        self._ws = {}
        self._step_seed = 0
        self.grad_ready_hook = None    # callable(flat, start, end) fired as gradient ranges become final
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------ parameters
    def _init_weights(self, seed):
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        std = self.config.initializer_range
        for n in self._flat.names:
            v = self._flat.view(n)
            if n.endswith("LayerNorm.weight"):
                v.fill_(1.0)
            elif n.endswith(".bias"):
                v.zero_()
            else:
                v.copy_(torch.randn(v.shape, generator=g) * std)

    def _rebind(self):
        named = dict(self.named_parameters())
        for n in self._flat.names:
            p = named[n]
            p.data = self._flat.view(n)
            p.grad = None

    def _apply(self, fn, recurse=True):
        self._flat.apply_(fn)
        self._anchor = fn(self._anchor.detach()).requires_grad_(True)
        self._rebind()
        self._ws = {}
        self.__dict__.pop("_err", None)
        return self

    def _bind_grads(self):
        self._flat.ensure_device_state()
        named = dict(self.named_parameters())
        for n in self._flat.names:
            p = named[n]
            if p.grad is None or p.grad.data_ptr() != self._flat.gview(n).data_ptr():
                p.grad = self._flat.gview(n)

    def flat_buffers(self):
        """[FlatBuffer] -- what the fused optimiser and the data-parallel gradient exchange operate on."""
        return [self._flat]

    def sync_weights(self):
        self._flat.sync_shadow(force=True)

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._flat._shadow_version = None

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_ws"] = {}
        st["grad_ready_hook"] = None
        st.pop("_err", None)
        return st

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, **_):
        """HF BertModel.forward (modeling_bert.py:623-686) -> namespace with ``pooler_output``.  ``position_ids`` ([B,S], [1,S] or
        [S]; None = arange(S)) are forwarded as nlp_classifier.py:23-27 / transformer_emb.py:20-24 forward them."""
        if not input_ids.is_cuda:
            raise MmsimError("BertModel.forward: inputs must be on the GPU; the HIP path has no CPU fallback")
        if position_ids is not None:
            if position_ids.dim() == 1:
                position_ids = position_ids.unsqueeze(0)
            position_ids = position_ids.to(device=input_ids.device, dtype=torch.long).expand(input_ids.shape).contiguous()
        # The fused attention kernels hold a whole sequence of 32 / 64 / 128 keys per workgroup.  Other lengths up to 128 (an
        # inference caller's max_length; the training path pads to 128, multimodal_dataset.py:44-48) run as the next supported
        # length with the extra positions masked out: masked keys get zero attention weight and only the [CLS] row is pooled,
        # so the result is that of the unpadded sequence.
        S = input_ids.shape[1]
        S_run = 32 if S <= 32 else (64 if S <= 64 else 128)
        if S > 128:
            raise ValueError(f"BertModel: sequences longer than 128 tokens are not supported by the fused attention kernels (got {S})")
        if S_run != S:
            if S_run > self.config.max_position_embeddings and position_ids is None:
                raise ValueError("BertModel: position table shorter than the padded sequence")
            pad = S_run - S
            F = torch.nn.functional
            if attention_mask is None:
                attention_mask = torch.ones_like(input_ids)
            input_ids = F.pad(input_ids, (0, pad))
            attention_mask = F.pad(attention_mask, (0, pad))
            token_type_ids = None if token_type_ids is None else F.pad(token_type_ids, (0, pad))
            position_ids = None if position_ids is None else F.pad(position_ids, (0, pad))
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if need_grad:
            pooled = _BertFn.apply(self._anchor, self, input_ids, token_type_ids, attention_mask, position_ids)
        else:
            ws = self._run_forward(input_ids, token_type_ids, attention_mask, keep=False, pids=position_ids)
            pooled = ws.pooled.clone()
        return SimpleNamespace(pooler_output=pooled, last_hidden_state=None)

    def _err_flag(self):
        f = self.__dict__.get("_err")
        dev = self._flat.master.device
        if f is None or f.device != dev:
            f = torch.zeros(1, dtype=torch.int32, device=dev)
            self.__dict__["_err"] = f
            ops.register_error_flag(f, IndexError, "BertModel: a token / token-type / position index was outside its embedding table")
        return f

    def check_indices(self):
        """Raise IndexError if any forward since the last check saw an index outside its embedding table (nn.Embedding raises
        eagerly; here the kernels clamp the index and raise a device flag, read on request: one host sync)."""
        f = self.__dict__.get("_err")
        if f is not None and f.is_cuda and int(f.item()) != 0:
            f.zero_()
            raise IndexError("BertModel: a token / token-type / position index was outside its embedding table")

    def _workspace(self, B, S, keep):
        key = (B, S, keep)
        ws = self._ws.get(key)
        if ws is None:
            ws = _Workspace(self.config, B, S, self._flat.master.device, keep)
            self._ws[key] = ws
        return ws

    def _run_forward(self, ids, tts, mask, keep, pids=None):
        cfg, fl = self.config, self._flat
        if ids.dim() != 2:
            raise ValueError("input_ids must be [B, S]")
        B, S = ids.shape
        H, L, nh = cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads
        if pids is None and S > cfg.max_position_embeddings:
            raise ValueError("sequence longer than max_position_embeddings")
        fl.sync_shadow()
        ws = self._workspace(B, S, keep)
        ws.gen += 1
        train = self.training
        ph = cfg.hidden_dropout_prob if train else 0.0
        pa = cfg.attention_probs_dropout_prob if train else 0.0
        if train:
            self._step_seed += 1
        seed = (self._step_seed * 0x9E3779B97F4A7C15 + 0x1234567) & 0xFFFFFFFFFFFFFFFF
        ids = ids.contiguous()
        tts = tts.contiguous() if tts is not None else None
        mask = mask.contiguous() if mask is not None else None
        ws.meta = SimpleNamespace(B=B, S=S, ids=ids, tts=tts, mask=mask, ph=ph, pa=pa, seed=seed, pids=pids)
        V = fl.view
        ops.embed_ln_fwd(ids, tts, V("embeddings.word_embeddings.weight"), V("embeddings.position_embeddings.weight"),
                         V("embeddings.token_type_embeddings.weight"), V("embeddings.LayerNorm.weight"),
                         V("embeddings.LayerNorm.bias"), ws.h[0], B, S, H, cfg.layer_norm_eps, self._err_flag(), ph, seed, 0,
                         pids=pids)
        I = cfg.intermediate_size
        for li in range(L):
            k = li if keep else 0
            p = f"encoder.layer.{li}."
            x = ws.h[k]
            wqkv = fl.sview(p + "attention.self.query.weight", (3 * H, H))
            bqkv = fl._view(fl.master, p + "attention.self.query.bias", (3 * H,))
            ops.gemm(x, wqkv, ws.qkv[k], bias=bqkv)
            ops.attn_fwd(ws.qkv[k], mask, ws.ctx[k], ws.lse[k], B, S, nh, H, pa, seed, 4 * li + 3)
            ops.gemm(ws.ctx[k], fl.sview(p + "attention.output.dense.weight"), ws.t, bias=V(p + "attention.output.dense.bias"))
            st = ws.st[k]
            ops.add_ln_fwd(ws.t, x, V(p + "attention.output.LayerNorm.weight"), V(p + "attention.output.LayerNorm.bias"),
                           ws.y1[k], ws.h1[k], st[0], st[1], cfg.layer_norm_eps, ph, seed, 4 * li + 1)
            ops.gemm(ws.h1[k], fl.sview(p + "intermediate.dense.weight"), ws.u[k], bias=V(p + "intermediate.dense.bias"),
                     epilogue=ops.EPI_GELU_DGELU if _GELU_PAIR else ops.EPI_GELU, aux_out=ws.upre[k])       # upre holds gelu'(pre-activation): the dgrad epilogue is one multiply
            ops.gemm(ws.u[k], fl.sview(p + "output.dense.weight"), ws.t, bias=V(p + "output.dense.bias"))
            ops.add_ln_fwd(ws.t, ws.h1[k], V(p + "output.LayerNorm.weight"), V(p + "output.LayerNorm.bias"),
                           ws.y2[k], ws.h[k + 1] if keep else ws.h[0], st[2], st[3], cfg.layer_norm_eps, ph, seed,
                           4 * li + 2)
        hl = ws.h[L] if keep else ws.h[0]
        ops.lib.gather_cls(hl.data_ptr(), ws.cls.data_ptr(), B, S, H, ops._stream())
        ops.gemm(ws.cls, fl.sview("pooler.dense.weight"), ws.pooled, bias=V("pooler.dense.bias"), epilogue=ops.EPI_TANH)
        return ws

    # ------------------------------------------------------------------ backward
    def _run_backward(self, ws, dpooled):
        cfg, fl = self.config, self._flat
        m = ws.meta
        B, S = m.B, m.S
        H, L, nh, I = cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.intermediate_size
        M = B * S
        self._bind_grads()
        V, G, SV = fl.view, fl.gview, fl.sview
        stream = ops._stream()
        dpooled = dpooled.contiguous().float()
        ops.lib.tanh_bwd(dpooled.data_ptr(), ws.pooled.data_ptr(), ws.dpre.data_ptr(), B * H, stream)
        ops.gemm(ws.dpre, ws.cls, G("pooler.dense.weight"), trans_a=True, b_kmajor=False, accumulate=True)
        ops.colsum(ws.dpre, G("pooler.dense.bias"))
        ops.gemm(ws.dpre, SV("pooler.dense.weight"), ws.dcls, b_kmajor=False)
        dh = ws.dh[0]
        ops.lib.scatter_cls(ws.dcls.data_ptr(), dh.data_ptr(), B, S, H, stream)
        if self.grad_ready_hook:
            self.grad_ready_hook(fl, *fl.span("pooler.dense.weight", "pooler.dense.bias"))
        skH = ops.pick_split_k(H, H, M)
        skI = ops.pick_split_k(I, H, M)
        sk3 = ops.pick_split_k(3 * H, H, M)
        pair = _WGRAD_PAIR and ops.wgrad_pair_eligible(3 * H, H, H, M)
        sk_pair = ops.pick_split_k(4 * H, H, M) if pair else 1
        # MMSIM_WGRAD_STREAM=1: the weight-gradient products (nothing in the backward pass depends on them) on their own HIP stream.
        # The data-gradient chain then never waits behind them, and -- the point -- two GEMM launches with different tile durations
        # share the chip, so that one launch's epilogue write burst meets the other's main loop instead of 256 CUs storing in lockstep.
        ws_on = _WGRAD_STREAM and dh.is_cuda
        if ws_on:
            if getattr(ws, "wstream", None) is None:
                ws.wstream = torch.cuda.Stream(device=dh.device)
                ops.register_side_stream(ws.wstream)
            main = torch.cuda.current_stream()
            wstream = ws.wstream
            pending = {}          # buffer name -> event of the weight-gradient launch that still reads it

        def on_wstream(reads, fn):
            """fn() on the weight-gradient stream, behind everything enqueued on the main stream so far; `reads`: scratch buffers whose
            next writer (on the main stream) has to wait for it."""
            if not ws_on:
                fn()
                return
            ev = torch.cuda.Event()
            ev.record(main)
            wstream.wait_event(ev)
            with torch.cuda.stream(wstream):
                fn()
                done = torch.cuda.Event()
                done.record(wstream)
            for r in reads:
                pending[r] = done

        def before_write(*names):
            if ws_on:
                for r in names:
                    ev = pending.pop(r, None)
                    if ev is not None:
                        main.wait_event(ev)

        for li in range(L - 1, -1, -1):
            p = f"encoder.layer.{li}."
            st = ws.st[li]
            # ---- output LayerNorm + FFN
            dy2 = ws.dy[0]
            dtA = ws.dt
            dT = dtA if m.ph > 0 else dy2
            before_write("dy0", "dtA")
            ops.ln_bwd(dh, None, ws.y2[li], st[2], st[3], V(p + "output.LayerNorm.weight"), dy2, dtA if m.ph > 0 else None,
                       G(p + "output.LayerNorm.weight"), G(p + "output.LayerNorm.bias"), G(p + "output.dense.bias"),
                       m.ph, m.seed, 4 * li + 2)
            on_wstream(("dtA", "dy0"), lambda dT=dT: ops.gemm(dT, ws.u[li], G(p + "output.dense.weight"), trans_a=True, b_kmajor=False,
                                                              split_k=skI, accumulate=True))
            before_write("du")
            ops.gemm(dT, SV(p + "output.dense.weight"), ws.du, b_kmajor=False, epilogue=ops.EPI_MUL if _GELU_PAIR else ops.EPI_MUL_GELU_GRAD,
                     aux_in=ws.upre[li])
            # dW and the bias gradient of intermediate.dense from ONE pass over du (the column sums ride on the weight-gradient MFMAs)
            on_wstream(("du",), lambda: ops.gemm_wgrad_colsum(ws.du, ws.h1[li], G(p + "intermediate.dense.weight"),
                                                             G(p + "intermediate.dense.bias"), skI))
            ops.gemm(ws.du, SV(p + "intermediate.dense.weight"), ws.dhb, b_kmajor=False)
            # ---- attention-output LayerNorm: dh1 = dy2 (residual) + dhb
            dy1 = ws.dy[1]
            dtB = ws.dt2 if ws_on else ws.dt
            dT = dtB if m.ph > 0 else dy1
            before_write("dy1", "dtB")
            ops.ln_bwd(dy2, ws.dhb, ws.y1[li], st[0], st[1], V(p + "attention.output.LayerNorm.weight"), dy1,
                       dtB if m.ph > 0 else None, G(p + "attention.output.LayerNorm.weight"),
                       G(p + "attention.output.LayerNorm.bias"), G(p + "attention.output.dense.bias"), m.ph, m.seed,
                       4 * li + 1)
            dT_o = dT
            if not pair:
                on_wstream(("dtB", "dy1"), lambda dT=dT: ops.gemm(dT, ws.ctx[li], G(p + "attention.output.dense.weight"), trans_a=True,
                                                                  b_kmajor=False, split_k=skH, accumulate=True))
            ops.gemm(dT, SV(p + "attention.output.dense.weight"), ws.dctx, b_kmajor=False)
            # ---- attention
            qkv_db = fl._view(fl.grad, p + "attention.self.query.bias", (3 * H,))
            before_write("dqkv")
            ops.attn_bwd(ws.qkv[li], m.mask, ws.ctx[li], ws.dctx, ws.lse[li], ws.dqkv, B, S, nh, H, m.pa, m.seed, 4 * li + 3,
                         dbias=qkv_db if _FUSE_ATTN_BIAS else None)                               # + the q|k|v bias gradients
            if not _FUSE_ATTN_BIAS:
                ops.colsum(ws.dqkv, qkv_db)
            if pair:      # q|k|v (48 tiles) and attention-output (16 tiles) weight gradients as one 256-block launch
                on_wstream(("dqkv", "dtB", "dy1"), lambda: ops.gemm_wgrad_pair(
                    ws.dqkv, ws.h[li], fl._view(fl.grad, p + "attention.self.query.weight", (3 * H, H)),
                    dT_o, ws.ctx[li], G(p + "attention.output.dense.weight"), sk_pair))
            else:
                on_wstream(("dqkv",), lambda: ops.gemm(ws.dqkv, ws.h[li], fl._view(fl.grad, p + "attention.self.query.weight", (3 * H, H)),
                                                      trans_a=True, b_kmajor=False, split_k=sk3, accumulate=True))
            nxt = ws.dh[1] if dh is ws.dh[0] else ws.dh[0]
            ops.gemm(ws.dqkv, fl.sview(p + "attention.self.query.weight", (3 * H, H)), nxt, b_kmajor=False,
                     epilogue=ops.EPI_ADD, aux_in=dy1)
            dh = nxt
            if self.grad_ready_hook:
                if ws_on:
                    main.wait_stream(wstream)          # the layer's weight gradients are final before their range is handed to the exchange
                    pending.clear()
                self.grad_ready_hook(fl, *fl.span(p + "attention.self.query.weight", p + "output.LayerNorm.bias"))
        if ws_on:
            main.wait_stream(wstream)                  # every weight gradient is in the flat buffer before the optimiser / exchange reads it
        ops.embed_ln_bwd(dh, m.ids, m.tts, V("embeddings.word_embeddings.weight"), V("embeddings.position_embeddings.weight"),
                         V("embeddings.token_type_embeddings.weight"), V("embeddings.LayerNorm.weight"),
                         G("embeddings.word_embeddings.weight"), G("embeddings.position_embeddings.weight"),
                         G("embeddings.token_type_embeddings.weight"), G("embeddings.LayerNorm.weight"),
                         G("embeddings.LayerNorm.bias"), B, S, H, cfg.layer_norm_eps, self._err_flag(), m.ph, m.seed, 0,
                         pids=m.pids)
        if self.grad_ready_hook:
            self.grad_ready_hook(fl, *fl.span("embeddings.word_embeddings.weight", "embeddings.LayerNorm.bias"))


class _BertFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, ids, tts, mask, pids=None):
        ws = model._run_forward(ids, tts, mask, keep=True, pids=pids)
        ctx.model, ctx.ws, ctx.gen = model, ws, ws.gen
        return ws.pooled.clone()

    @staticmethod
    def backward(ctx, dpooled):
        # the saved activations live in the module's per-(B, S) workspace, which the next grad-enabled forward of the same
        # shape overwrites: backward of an older forward would silently use the newer activations
        if ctx.ws.gen != ctx.gen:
            raise MmsimError("BertModel: backward of a forward whose activation workspace has been overwritten by a later "
                             "forward of the same (B, S); run backward before the next grad-enabled forward of this shape")
        ctx.model._run_backward(ctx.ws, dpooled)
        return None, None, None, None, None, None


def as_native(ptm):
    """Return ``ptm`` if it already is the native tower, else convert an HF-style BERT module (weights copied)."""
    if isinstance(ptm, BertModel):
        return ptm
    cfg = getattr(ptm, "config", None)
    if cfg is None or not hasattr(ptm, "state_dict"):
        raise TypeError("pretrained_model must be a multimodalsimilar_amd BertModel or an HF-style BERT module")
    native = BertModel(BertConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
        max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
        layer_norm_eps=cfg.layer_norm_eps, hidden_dropout_prob=cfg.hidden_dropout_prob,
        attention_probs_dropout_prob=cfg.attention_probs_dropout_prob))
    sd = {k: v for k, v in ptm.state_dict().items() if "position_ids" not in k}
    native.load_state_dict(sd, strict=True)
    dev = next(ptm.parameters()).device
    if dev.type != "cpu":
        native.to(dev)
    return native
