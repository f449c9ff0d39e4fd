import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, oracle_lib as o, provekit_amd
from provekit_amd._lib import lib
from provekit_amd.field import random_field
ctx=provekit_amd.Context(0)
n=2000
a=random_field(n,1); a[:4]=o.ints_to_limbs([0,1,o.P-1,o.P-2]); b=random_field(n,2)
da=ctx.upload(a); db=ctx.upload(b); do=ctx.alloc_fe(n)
for op in range(14):
    h=np.empty_like(a)
    lib.pk_selftest_arith(op,a.ctypes.data,b.ctypes.data,h.ctypes.data,n)
    ctx._check(lib.pk_selftest_arith_device(ctx.handle,op,da.ptr,db.ptr,do.ptr,n))
    d=ctx.download_fe(do,n)
    bad=np.nonzero((d!=h).any(axis=1))[0]
    print('op',op,'mismatch',len(bad),bad[:4])
    for i in bad[:2]:
        print('  a  ',hex(o.limbs_to_ints(a[i])[0])); print('  dev',hex(o.limbs_to_ints(d[i])[0])); print('  hst',hex(o.limbs_to_ints(h[i])[0]))
