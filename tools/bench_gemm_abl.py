"""Ablation of the pipelined GEMM main loop (GPU box): MMSIM_GEMM_DBG bits 1 = no DMA, 4 = no MFMA, 8 = no epilogue.
The switches are compiled only into a SEPARATE ablation library (-DMMSIM_ABLATE, built here into gpurun_out/); the product
library multimodalsimilar_amd/libmmsim_hip.so has no such code path and never reads the variable."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodalsimilar_amd import build as _b
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
ABL = os.path.join(ROOT, "gpurun_out", "libmmsim_hip_ablate.so")
srcs = [os.path.join(_b.CSRC, f) for f in _b.SOURCES]
subprocess.check_call([_b.HIPCC] + _b.FLAGS + ["-DMMSIM_ABLATE", "-shared", "-o", ABL] + srcs)
os.environ["MMSIM_LIB"] = ABL
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from multimodalsimilar_amd import ops
M=32768
shapes=(("o-NT",1024,1024,"nt"),("ffn1-NT",4096,1024,"nt"),("ffn2-NT",1024,4096,"nt"),("ffn1-NN",1024,4096,"nn"),
        ("qkv-TN",3072,1024,"tn"),("o-TN",1024,1024,"tn"),("ffn1-TN",4096,1024,"tn"),("ffn2-TN",1024,4096,"tn"))
only=os.environ.get("ABL_ONLY")
for name,N,K,lay in shapes:
    if only and only not in name: continue
    if lay=="tn":      # the weight gradient dW[N,K] += dy[M,N]^T x[M,K]: split-K, fp32 output by atomics
        dy=torch.randn(M,N,device="cuda").bfloat16(); x=torch.randn(M,K,device="cuda").bfloat16()
        dw=torch.zeros(N,K,device="cuda"); sk=ops.pick_split_k(N,K,M)
        f=lambda: ops.gemm(dy,x,dw,trans_a=True,b_kmajor=False,split_k=sk,accumulate=True)
        f(); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/10
        print(f"  {name} (sk={sk}): {t*1e3:8.1f} us  ({2.0*M*N*K/t/1e9:7.1f} TF-equivalent)")
        continue
    if lay=="nt":
        a=torch.randn(M,K,device="cuda").bfloat16(); b=(torch.randn(N,K,device="cuda")*0.05).bfloat16(); kw={}
    else:
        a=torch.randn(M,K,device="cuda").bfloat16(); b=(torch.randn(K,N,device="cuda")*0.05).bfloat16(); kw=dict(b_kmajor=False)
    c=torch.empty(M,N,dtype=torch.bfloat16,device="cuda")
    f=lambda: ops.gemm(a,b,c,**kw)
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/10
    print(f"  {name}: {t*1e3:8.1f} us  ({2.0*M*N*K/t/1e9:7.1f} TF-equivalent)")
'''
modes = ((0, "full"), (8, "no epilogue"), (1, "no DMA"), (4, "no MFMA"), (9, "no DMA, no epilogue"), (12, "no MFMA, no epilogue"))
if os.environ.get("ABL_MODES"):
    want = [int(v) for v in os.environ["ABL_MODES"].split(",")]
    modes = tuple(m for m in modes if m[0] in want)
for dbg, label in modes:
    print(f"dbg={dbg} [{label}]", flush=True)
    env = dict(os.environ, MMSIM_GEMM_DBG=str(dbg))
    subprocess.run([sys.executable, "-c", code], env=env)
