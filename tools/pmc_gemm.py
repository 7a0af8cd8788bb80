"""One launch of each text-tower forward GEMM (for rocprofv3 --pmc runs): prints nothing, just launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
M = 32768
for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); b = torch.zeros(N, device="cuda")
    for _ in range(3):
        ops.gemm(x, w, y, bias=b)
torch.cuda.synchronize()
