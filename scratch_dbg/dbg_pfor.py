import sys; sys.path.insert(0,'/root/repo')
import numpy as np, btrblocks_amd as B
from oracle import pyoracle
sys.path.insert(0,'/root/repo/tests')
import cases
for name, v, nm, spec, kw in cases.cases():
    if name!="int_pfor_outliers": continue
    c=B.compress_chunk(v,nm,spec,**kw)
    ctx=B.GpuContext(0)
    got=ctx.decode_chunk_host(c)[0].view(np.uint32) if isinstance(ctx.decode_chunk_host(c),tuple) else ctx.decode_chunk_host(c).view(np.uint32)
    want=v.view(np.uint32)
    bad=np.nonzero(got!=want)[0]
    print(len(bad), bad[:40])
    for i in bad[:20]: print(i, i%256, hex(got[i]), hex(want[i]), int(want[i]).bit_length())
    allexc=np.nonzero(want>=1024)[0]
    print("exceptions", len(allexc), "bad among first block ids", sorted(set(bad//256))[:30])
