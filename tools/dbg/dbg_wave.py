import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
from cases import WINDOW_SETS
from nnmnkwii_amd import _hip
for wn in ('static', 'std2', 'std3'):
  w = WINDOW_SETS[wn]; nw = len(w)
  for T in (1025,):
    for sd in (12,):
      rng = np.random.RandomState(0)
      m = torch.from_numpy(rng.randn(1, T, nw*sd)).cuda(); v = torch.from_numpy(rng.rand(1, T, nw*sd)+0.1).cuda()
      yw,_ = _hip.forward(m, v, w, None, algo=2); yg,_ = _hip.forward(m, v, w, None, algo=1)
      e = (yw-yg).abs()[0]
      badt = (e.max(dim=1).values > 1e-9).nonzero().flatten().tolist()
      print(wn, T, sd, 'maxerr', float(e.max()), 'nbad frames', len(badt), badt[:40])
