"""Dump and summarise the per-wave timeline of ONE tile-10 GEMM launch (PSAM_GEMM_TRACE=<file> makes the library write
[block][wave][8] u64: t_entry, t_loop_start, t_loop_end, t_stores_issued, t_stores_acked (100 MHz ticks), HW_ID, XCC_ID,
block).   python tools/gemm_trace.py [M N K] [f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = "/tmp/gemm_trace.bin"
import numpy as np, torch
from protosam_amd import ops
dev = torch.device("cuda:0")
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (65536, 3840, 1280)
f32 = len(sys.argv) > 4 and sys.argv[4] == "f32"
a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.float16)
bias = torch.randn(N, device=dev); gamma = torch.randn(N, device=dev)
ops.gemm_set_tile(10)
def run():
    if f32: ops.gemm(a, w, bias, out=out, epilogue=ops.EPI_F32, resid=out, gamma=gamma)
    else: ops.gemm(a, w, None, out=out, epilogue=ops.EPI_F16)
for _ in range(3): run()
torch.cuda.synchronize()
d = np.fromfile(path, dtype=np.uint64).reshape(-1, 8, 8).astype(np.int64)
d = d[d[:, 0, 0] > 0]
t0 = d[:, :, 0].min()
T = (d[:, :, :5] - t0) * 0.01          # us
hw = d[:, 0, 5]; xcc = d[:, 0, 6] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 10 + cu * 1
cyc = d[:, :, 7].astype(np.float64); dur = (d[:, :, 4] - d[:, :, 0]) * 0.01
print(f"shader clock over the workgroups' lifetimes: {np.median(cyc / np.maximum(dur, 1e-9)) / 1e3:.3f} GHz (s_memtime / wall)")
print("blocks", len(d), "distinct CUs", len(set(cuid.tolist())), "kernel span us", T[:, :, 4].max())
ent = T[:, :, 0].min(1); ls = T[:, :, 1].max(1); le = T[:, :, 2].max(1); si = T[:, :, 3].max(1); sa = T[:, :, 4].max(1)
print(f"per block (us): prologue {np.mean(ls-ent):.2f}  k-loop {np.mean(le-ls):.2f}  epilogue issue {np.mean(si-le):.2f}  "
      f"store ack {np.mean(sa-si):.2f}  total {np.mean(sa-ent):.2f}")
print(f"  percentiles of epilogue issue: {np.percentile(si-le,[5,50,95])}   ack: {np.percentile(sa-si,[5,50,95])}")
# gaps between consecutive blocks on the same CU
gaps = []
for c in set(cuid.tolist()):
    idx = np.where(cuid == c)[0]
    o = idx[np.argsort(ent[idx])]
    for i in range(1, len(o)):
        gaps.append(ent[o[i]] - sa[o[i - 1]])
gaps = np.array(gaps)
print(f"gap (last ack -> next block entry on the same CU): mean {gaps.mean():.2f} us, pct {np.percentile(gaps,[5,50,95])}")
first = np.array([ent[np.where(cuid == c)[0]].min() for c in set(cuid.tolist())])
print(f"first-round entry spread: {first.min():.2f} .. {first.max():.2f} us")
for r in range(3):
    sel = np.argsort(ent)[r*256:(r+1)*256]
    print(f"round {r}: entry {ent[sel].min():.1f}..{ent[sel].max():.1f}  loop end {le[sel].min():.1f}..{le[sel].max():.1f}  acked {sa[sel].min():.1f}..{sa[sel].max():.1f}")
