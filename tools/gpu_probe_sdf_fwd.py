"""GPU probe (dev tool): the fused SDF-query forward alone, for PMC passes:
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d out -- python3 tools/gpu_probe_sdf_fwd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch, torch.nn as nn
from d3h import sdf_mlp
torch.manual_seed(0)
dims = [(39, 256), (256, 256), (256, 256), (256, 256), (295, 256), (256, 256), (256, 256), (256, 1)]
params = []
for i, o in dims:
    l = nn.Linear(i, o); params += [l.weight.detach().cuda(), l.bias.detach().cuda()]
n = 262144
x = (torch.rand(n, 3, device='cuda') * 2.4 - 1.2)
wp = sdf_mlp.pack_weights({k: p for k, p in zip(sdf_mlp._PARAM_ORDER, params)}, prefix='')
save = len(sys.argv) > 1 and sys.argv[1] == 'save'
for _ in range(6):
    sdf_mlp.forward(x, wp, save=save)
torch.cuda.synchronize()
