import os, sys, numpy as np, torch
sys.path.insert(0, 's-blas_amd/python')
os.environ['SBLAS_SPMM_VARIANT'] = os.environ.get('SBLAS_SPMM_VARIANT', 'win4')
os.environ['SBLAS_ABLATE'] = '4'
import sblas_amd as S
from sblas_amd import synth
dev = torch.device('cuda:0')
rows, (rp, ci, v) = synth.nd24k_like()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
n = 64
rowptr, colidx, val = d(rp), d(ci), d(v)
B = torch.rand(rows*n, dtype=torch.float64, device=dev); C = torch.ones(rows*n, dtype=torch.float64, device=dev)
ws = torch.empty(S.spmm_workspace_bytes(rows, rows, len(ci), n)//8, dtype=torch.float64, device=dev)
for it in range(3):
    S.cycle_stamps()
    S.spmm(rows, rows, rowptr, colidx, val, B, rows, n, 1.0, 1.0, C, rows, ws)
    torch.cuda.synchronize()
    g = S.cycle_stamps()
nc, nl = g[7], g[8]
print('consumer waves', nc, 'loader waves', nl, '(per-wave averages in shader cycles, per panel)')
print('consumer: prologue %.0f  visits %.0f  barrier-wait %.0f' % (g[0]/nc, g[1]/nc, g[2]/nc))
print('loader  : put|dma-issue %.0f  fetch-issue|dma-wait %.0f  barrier-wait %.0f' % (g[4]/nl, g[5]/nl, g[6]/nl))
print('whole-wave avg %.0f' % (g[9]/(nc+nl)))
if os.environ['SBLAS_SPMM_VARIANT'] == 'win4':
    tiles = g[11]/nc
    print('gen 4: tiles/panel %.1f; per tile: visits %.0f (window wait %.0f, select+16 slots %.0f), barrier %.0f'
          % (tiles, g[1]/nc/tiles, g[3]/nc/tiles, g[10]/nc/tiles, g[2]/nc/tiles))
    print('gen 4 loader per tile: dma issue %.0f, dma landing %.0f, barrier %.0f'
          % (g[4]/nl/tiles, g[5]/nl/tiles, g[6]/nl/tiles))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(5):
    S.spmm(rows, rows, rowptr, colidx, val, B, rows, n, 1.0, 1.0, C, rows, ws)
e1.record(); torch.cuda.synchronize()
print('stamped call: %.1f us' % (e0.elapsed_time(e1) * 200))
