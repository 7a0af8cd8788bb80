"""Is the pipelined GEMM's epilogue bound per CU or by the chip?  One round of 256 x 256 tiles with 32 / 64 / 128 / 256 tiles (workgroups) on the chip,
K = 1024, forward layout with bias, full kernel against the ablation build's "no epilogue" (MMSIM_GEMM_DBG=8): the difference is the epilogue's
exposed time at that number of concurrently storing CUs.  Needs gpurun_out/libmmsim_hip_ablate.so (tools/bench_gemm_abl.py builds it)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.path.join(ROOT, "gpurun_out", "libmmsim_hip_ablate.so")
if not os.path.exists(ABL):
    sys.path.insert(0, ROOT)
    from multimodalsimilar_amd import build as _b
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    subprocess.check_call([_b.HIPCC] + _b.FLAGS + ["-DMMSIM_ABLATE", "-shared", "-o", ABL] + [os.path.join(_b.CSRC, f) for f in _b.SOURCES])
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from multimodalsimilar_amd import ops
N, K = 1024, 1024
for tiles in (32, 64, 128, 256, 512):
    M = tiles // 4 * 256
    a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); bias = torch.zeros(N, device="cuda")
    aux = torch.empty_like(c)
    for name, kw in (("bias", dict(bias=bias)), ("gelu-pair", dict(bias=bias, epilogue=ops.EPI_GELU_DGELU, aux_out=aux))):
        f = lambda: ops.gemm(a, b, c, **kw)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        print(f"  tiles={tiles:4d} {name:9s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
'''
for dbg, label in ((0, "full"), (8, "no epilogue")):
    print(f"dbg={dbg} [{label}]", flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MMSIM_LIB=ABL, MMSIM_GEMM_DBG=str(dbg)), cwd=ROOT)
