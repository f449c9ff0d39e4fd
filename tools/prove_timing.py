import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch; torch.cuda.is_available()
import numpy as np, provekit_amd, oracle_lib as oracle
from test_gpu_prove import size_class_instance
from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
from provekit_amd.sparse_matrix import R1CS, SparseMatrix
m=21
ctx=provekit_amd.Context(0)
nc,nw,mats,interner,z=size_class_instance(oracle,m)
r1cs=R1CS(ctx,*(SparseMatrix(nc,nw,*t) for t in mats),interner)
s=WhirR1CSScheme(ctx,r1cs,m,m-1,WhirConfig.derive(m),blinding_config_for(m-1))
dz=ctx.upload(z)
for i in range(3): s.prove_nocopy(dz,seed=i)
t=time.perf_counter()
for i in range(10): s.prove_nocopy(dz,seed=10+i)
print("ms/proof", (time.perf_counter()-t)*100)
os.environ["PK_PROVE_TIMING"]="1"
s.prove_nocopy(dz,seed=99)
