import os, sys, torch
sys.path.insert(0, os.getcwd())
from humanvid_amd import lib as hvlib, ops
BF16=torch.bfloat16
dev=hvlib.require_gpu(); L=hvlib.load(); st=hvlib.current_stream()
def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(iters):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return ts
N_tok, C, D = 6144, 320, 40
for n, mode in [(48,'mixed'),(24,'cond'),(24,'uncond'),(48,'cond'),(24,'cond_sepbuf'),(24,'cond_ldpad')]:
    M=n*N_tok
    ldq = 2*C if mode!='cond_sepbuf' else 3*C
    qk=(torch.randn(M, ldq, device=dev)).to(BF16)
    ldvt = M if mode!='cond_ldpad' else M+64
    vt=torch.randn(C, ldvt, device=dev).to(BF16)
    k2, vt2 = torch.randn(2*N_tok, C, device=dev).to(BF16), torch.randn(C, 2*N_tok, device=dev).to(BF16)
    o=torch.empty(M, C, dtype=BF16, device=dev)
    if mode=='mixed': sel=[-1]*(n//2)+[1]*(n-n//2)
    elif mode.startswith('cond'): sel=[1]*n
    else: sel=[-1]*n
    sel=torch.tensor(sel, dtype=torch.int32, device=dev)
    ts=timeit(lambda: ops.attention(L, st, qk, qk[:, C:], vt, o, n_images=n, heads=8, D=D, Lq=N_tok, L1=N_tok, ldq=ldq, ldk=ldq, ldvt=ldvt, ldo=C, k2=k2, vt2=vt2, ldk2=C, ldvt2=2*N_tok, L2=N_tok, bank_sel=sel))
    print(f"n={n:3d} {mode:12s} ms per launch:", ' '.join(f'{t:.3f}' for t in ts), flush=True)
