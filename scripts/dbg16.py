import sys,os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_binding as ob
import pogs_amd
from pogs_amd import graph as G
soa=lambda fv:{k:getattr(fv,k) for k in "habcde"}
rel=lambda a,b: float(np.linalg.norm(np.asarray(a,np.float64)-np.asarray(b,np.float64))/np.linalg.norm(np.asarray(b,np.float64)))
rng=np.random.default_rng(0)
A=rng.standard_normal((600,300)); Wz=rng.standard_normal((200,900))
r1=np.outer(rng.standard_normal(800), rng.standard_normal(250))
M=A*np.exp(rng.uniform(-8,8,(600,1)))*np.exp(rng.uniform(-8,8,(1,300)))
for m in (50,20,600,600,200,800,600,600): b=rng.standard_normal(m)
f,g=G.lasso_functions(b,0.1,300)
want=ob.oracle_solve(M,soa(f),soa(g),dtype=np.float32)
for env in ({}, {"POGS_AMD_FUSED":"0"}, {"POGS_AMD_SK_FULL":"1"}, {"POGS_AMD_GRAM":"fp32"}, {"POGS_AMD_DEFER":"0"}):
    os.environ.update(env)
    got=pogs_amd.graph._solve_graph_form(M,f,g,1e-4,1e-4,2500,0,1.0,dtype=np.float32)
    for k in env: os.environ.pop(k)
    print(env, got["status"], got["iterations"], want["iterations"], rel(got["x"],want["x"]), rel(got["y"],want["y"]))
