import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import bench
from med_ts_llm_amd.models.backbone import FrozenBackbone, random_state_dict
hf = dict(bench.WORKLOADS["llama2_7b_semseg_B32_L1024_C12"][0], num_hidden_layers=2)
sd = random_state_dict(hf, seed=0, std=0.02, device="cuda", dtype=torch.bfloat16)
bb = FrozenBackbone(hf, sd, torch.device("cuda"))
torch.manual_seed(0)
h0 = torch.randn(32, 256, 4096, device="cuda") * 0.1
dout = (torch.randn(32, 128, 4096, device="cuda") * 0.01).to(torch.bfloat16)
out, saved = bb.run_forward(h0, 128, keep=True, n_save=128)
dh0 = bb.run_backward(h0, dout, saved, 128, n_grad=128)
torch.save({"out": out.cpu(), "dh0": dh0[:, 128:].cpu()}, sys.argv[1])
