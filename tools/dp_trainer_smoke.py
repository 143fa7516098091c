"""2-rank trainer smoke on ONE GPU (gloo on device tensors): python -m torch.distributed.run --nproc-per-node 2 tools/dp_trainer_smoke.py
Trains one epoch with the row-sharded mapping layer, evaluates (stitched, unsharded), writes a checkpoint (collective
state_dict) and checks that both ranks hold identical replicated parameters and that the checkpoint has full shapes."""
import os, sys, json, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MTL_DIST_BACKEND", "gloo")
import numpy as np, torch, torch.distributed as dist
from pathlib import Path
from safetensors.torch import save_file
from helpers import hf_cfg, GOLDEN
from med_ts_llm_amd.models.backbone import random_state_dict
from med_ts_llm_amd.tasks import get_trainer
from med_ts_llm_amd.tasks.windows import register_series
from med_ts_llm_amd.utils import dict_to_object

rank = int(os.environ["RANK"])
d = Path(tempfile.gettempdir()) / "mtl_dp_smoke_llm"
if rank == 0:
    d.mkdir(exist_ok=True)
    cfg = hf_cfg("gpt2"); sd = random_state_dict(cfg, seed=5, std=0.05)
    (d / "config.json").write_text(json.dumps(cfg)); save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
    shutil.copy(GOLDEN / "tokenizer.json", d / "tokenizer.json")

def source(config, split):
    g = np.random.default_rng({"train": 1, "val": 2, "test": 3}[split]); n = 400; t = np.arange(n, dtype=np.float32)
    return {"data": np.stack([np.sin(t / 5), np.cos(t / 9), np.sin(t / 13)], -1).astype(np.float32) + 0.05 * g.standard_normal((n, 3)).astype(np.float32)}
register_series("series_dp", source)
out = Path(tempfile.gettempdir()) / "mtl_dp_smoke_logs"
config = {"DEBUG": False, "task": "forecasting", "model": "medtsllm", "history_len": 64, "pred_len": 16, "paths": {"logdir": str(out)},
          "data": {"dataset": "series_dp", "mode": "multivariate", "cols": "all", "normalize": True, "step": 8},
          "training": {"epochs": 1, "batch_size": 8, "optimizer": "adam", "learning_rate": 2e-3, "dropout": 0.0, "loss": "mse",
                       "eval_metric": "mse", "eval_metric_direction": "min", "shuffle": False},
          "tasks": {"segmentation": {"mode": "boundary-prediction"}},
          "models": {"timellm": {"d_model": 8, "d_ff": 64, "n_heads": 2, "num_tokens": 64, "covariate_mode": "concat",
                                 "embedding_downsample_mode": "linear", "patching": {"patch_len": 16, "stride": 8},
                                 "prompting": {"dataset": True, "task": True, "clip": False, "input_stats": True, "examples": False, "input_stats_dim": 0, "input_stats_select": "all"},
                                 "llm": {"enabled": True, "llm": str(d), "llm_layers": -1, "load_in_4bit": False, "load_in_8bit": False}}},
          "setup": {"seed": 0, "device": "auto", "dtype": "mixed", "num_workers": 0, "logger": "print", "quiet": True}}
from med_ts_llm_amd import parallel
parallel.init_from_env("cuda")
dist.barrier()
tr = get_trainer("dp-smoke", dict_to_object(config))
assert tr.world_size == 2 and tr.model.mapping_layer.weight.shape[0] == 32, tr.model.mapping_layer.weight.shape
tr.train()
scores = tr.test()
w = tr.model.output_projection.linear.weight.detach().float()
both = [torch.zeros_like(w) for _ in range(2)]; dist.all_gather(both, w)
assert torch.equal(both[0], both[1]), "replicated parameters diverged between ranks"
sc = torch.tensor([scores["test/mse"]], device="cuda", dtype=torch.float64); scs = [torch.zeros_like(sc) for _ in range(2)]; dist.all_gather(scs, sc)
assert torch.equal(scs[0], scs[1])
dist.barrier()
if rank == 0:
    ck = torch.load(out / "dp-smoke" / "checkpoints" / "latest.pt")
    assert ck["model"]["mapping_layer.weight"].shape[0] == 64, ck["model"]["mapping_layer.weight"].shape
    print("dp trainer smoke OK", scores)
dist.destroy_process_group()
