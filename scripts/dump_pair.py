"""Dump one bench pair (scans as generated on the GPU) + the device's first-iteration result restricted to one voxel.  usage: dump_pair.py K VOXEL"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icet_amd
from icet_amd import lidar_sim as ls
from oracle import pyoracle as po
K, V = int(sys.argv[1]), int(sys.argv[2]); T, P = 75, 24
s1, s2, _ = ls.make_batch_pair(K, device=torch.device("cuda", 0))
a, b = s1.T.cpu().numpy(), s2.T.cpu().numpy()
sph = po.c2s(b).astype(np.float64)
vox = (np.floor(sph[:, 2] / np.pi * P).astype(int) % P) * T + (np.floor(sph[:, 1] / (2 * np.pi) * T).astype(int) % T)
bb = b[vox == V]
ctx = icet_amd.Context(0)
g = ctx.solve(a, bb, 1, np.zeros(6), 24, 75, aux=True)
np.savez_compressed("gpurun_out/pair_%d_voxel_%d.npz" % (K, V), scan1=a, scan2=b, scan2_voxel=bb, htwh=g["aux"]["htwh"], htwdz=g["aux"]["htwdz"], X=g["X"],
                    n2_raw=g["aux"]["n2_raw"], n2_in=g["aux"]["n2_in"], mu1=g["aux"]["mu1"][V], sigma1=g["aux"]["sigma1"][V], evecs1=g["aux"]["evecs1"][V],
                    l_diag=g["aux"]["l_diag"][V], bounds=g["aux"]["cluster_bounds"][V], n1_raw=g["aux"]["n1_raw"][V])
print("saved", len(a), len(b), len(bb))
