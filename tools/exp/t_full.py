import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import _pkg; pkg = _pkg.load()
import numpy as np, torch
import sbm_oracle as oracle
from test_gpu_fullsize import *
from u96_slam_amd import synth
W,H,nd,wsz,n,uniq = 3840,2160,256,21,32,2
t=time.time(); L,R = synth.make_batch(100, uniq, W, H, nd); print('gen', time.time()-t)
t=time.time(); dL,dR = device_batch(torch, L, R, n//uniq); bm = make_engine(pkg, nd, wsz, **FULL); out = bm.compute_device(dL,dR).cpu().numpy(); print('compute', time.time()-t, out.shape, (out>=0).mean())
t=time.time(); per,total = crc_rows(out); print('crc', time.time()-t, len(set(per)))
t=time.time(); print(small_components_left(out[0],50,32,-16), 'cc', time.time()-t)
nosp = dict(FULL, speckle_window_size=0, speckle_range=0)
bm2 = make_engine(pkg, nd, wsz, **nosp)
got = bm2.compute_device(dL[:1], dR[:1]).cpu().numpy()[0]
print('removed by speckle', (out[0]!=got).sum(), 'small comps before speckle', small_components_left(got,50,32,-16))
t=time.time(); band_vs_oracle(pkg, oracle, L[0], R[0], nd, wsz, H//2-10, 48, got); print('band', time.time()-t)
