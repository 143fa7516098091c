#!/usr/bin/env python3
"""Boundary proof, build container only: the REFERENCE's own trainer (`tasks.get_trainer`, reference `BaseTask.__init__`,
`build_optimizer`, logger `save_state`, `from_run_id`) driven with `models.model_lookup["medtsllm"]` swapped for the build's class — the
one-line change INTEGRATION.md describes — through everything that runs without a GPU, i.e. up to the first forward.
Prints one JSON object; tests/test_boundary.py runs it in a subprocess (the reference's top-level module names — models, tasks, datasets,
utils — must not leak into the test process)."""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))
import make_golden as MG  # noqa: E402


def main():
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        MG.STUBS["toml.py"] = ("import json, tomli\n"
                               "def load(p):\n    try:\n        with open(p, 'rb') as f:\n            return tomli.load(f)\n"
                               "    except Exception:\n        with open(p) as f:\n            return json.load(f)\n"
                               "def dump(d, f):\n    json.dump(d, f)\n")
        MG.setup_imports(tmp)
        d = str(Path(tmp) / "llm_gpt2")
        os.makedirs(d)
        MG.make_backbone("gpt2", d, seed=100)
        MG.make_tokenizer(d)
        import datasets as ref_datasets
        import models as ref_models
        import tasks as ref_tasks
        import tasks.base as ref_base
        from utils import dict_to_object
        from datasets.base import BaseDataset, ForecastDataset
        from med_ts_llm_amd.models import model_lookup as ours
        ref_models.model_lookup["medtsllm"] = ours["medtsllm"]          # <- the integration: one registry entry
        assert ref_base.model_lookup is ref_models.model_lookup

        class SynthBase(BaseDataset):
            """synthetic multichannel physiological waveforms sampled at 125 Hz."""
            supported_tasks = ["forecasting"]

            def get_data(self, split=None):
                g = torch.Generator().manual_seed({"train": 11, "val": 12, "test": 13}[split or self.split])
                n = 64 + 16 + 8 * 15
                t = torch.arange(n, dtype=torch.float32)
                data = torch.stack([torch.sin(t / 5.0), torch.cos(t / 9.0), 0.01 * t], dim=-1) + 0.1 * torch.randn(n, 3, generator=g)
                return {"data": data.numpy().astype(np.float32)}

        ref_datasets.dataset_lookup["synthetic"] = {"forecasting": type("SynthForecast", (SynthBase, ForecastDataset), {"__doc__": SynthBase.__doc__})}
        logdir = Path(tmp) / "logs"
        cfgd = MG.base_config(d, "forecasting", 64, 16, "concat", "linear", MG.PROMPT_FULL, dtype="fp32")
        cfgd.update({
            "DEBUG": False, "paths": {"logdir": str(logdir)},
            "data": {"dataset": "synthetic", "mode": "multivariate", "cols": "all", "normalize": True, "step": 8},
            "training": {"epochs": 1, "batch_size": 4, "optimizer": "adamw", "learning_rate": 1e-3, "dropout": 0.1,
                         "loss": "mse", "eval_metric": "mse", "eval_metric_direction": "min"},
            "setup": {"seed": 0, "device": "cpu", "dtype": "fp32", "num_workers": 0, "logger": "print"},
            "datasets": {"synthetic": {}},
        })
        trainer = ref_tasks.get_trainer("boundary-probe", dict_to_object(cfgd))            # the REFERENCE's BaseTask.__init__ end to end
        model = trainer.model
        res["trainer_class"] = f"{type(trainer).__module__}.{type(trainer).__name__}"
        res["model_class"] = f"{type(model).__module__}.{type(model).__name__}"
        res["optimizer"] = type(trainer.optimizer).__name__
        res["optimizer_param_names"] = sorted(n for n, p in model.named_parameters() if any(p is q for g in trainer.optimizer.param_groups for q in g["params"]))
        res["trainable"] = sorted(n for n, p in model.named_parameters() if p.requires_grad)
        res["dtype_device"] = sorted({f"{p.dtype}/{p.device}" for p in model.parameters()})
        res["supported_tasks"] = model.supported_tasks
        # prepare_batch (reference) + the model's prompt builder on a reference batch
        batch = trainer.prepare_batch(next(iter(trainer.train_dataloader)))
        res["batch_keys"] = sorted(batch)
        res["prompt0"] = model.build_prompt(batch)[0]
        # checkpoint surface through the reference's logger and from_run_id
        trainer.logger.save_state("latest")
        ck = torch.load(logdir / "boundary-probe" / "checkpoints" / "latest.pt")
        res["checkpoint_keys"] = sorted(ck)
        res["checkpoint_model_keys"] = list(ck["model"])
        with torch.no_grad():
            model.mapping_layer.bias.add_(1.0)
        trainer.logger.save_state("best")
        again = type(trainer).from_run_id("boundary-probe", ckpt="best", basepath=str(logdir))
        res["from_run_id_restored"] = bool(torch.equal(again.model.mapping_layer.bias, model.mapping_layer.bias)) and type(again.model) is type(model)
        # the first forward is where the CPU ends: the product has no CPU path
        try:
            model(batch)
            res["forward_on_cpu"] = "ran"
        except RuntimeError as e:
            res["forward_on_cpu"] = str(e)
        # the reference's fine-tuning entry point
        res["load_pretrained_keys"] = model.load_pretrained({k: v.clone() for k, v in ck["model"].items()})
    print("BOUNDARY_PROBE " + json.dumps(res))


if __name__ == "__main__":
    main()
