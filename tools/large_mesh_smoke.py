#!/usr/bin/env python3
"""Newmark steps with the default (multigrid) solver on large meshes of the other element families: 2D Q1..Q4 and
3D Q1 (0.6-3 M dofs).  python tools/large_mesh_smoke.py"""
import sys, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
from bench import _pkg
M=_pkg()
for dim,p,reps,hi in ((2,2,(512,512),(1.0,1.0)),(2,3,(300,300),(1.0,1.0)),(2,1,(1024,1024),(1.0,1.0)),(2,4,(200,100),(2.0,1.0)),(3,1,(100,100,100),(1,1,1))):
    G=M.Context(dim=dim,degree=p,reps=reps,hi=hi)
    t=(0.0,-2e3,0.0)[:dim]
    for k in range(2):
        G.set_interface_traction(tuple(0.5*(k+1)*x for x in t))
        t0=time.perf_counter(); rc,info=G.newmark_step(tol_lin=1e-6,max_it_mult=1.0); dt=time.perf_counter()-t0
    print(dim,p,reps,G.n,"rc",rc,"newton",info.newton_iterations,"cg",info.lin_its_total,"%.1f ms"%(1e3*dt),flush=True)
    G.close()
