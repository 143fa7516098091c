"""SURVEY.md 8b in executable form (build container only — skipped where /root/reference does not exist, e.g. on the GPU box):
the reference's own `tasks.get_trainer` / `BaseTask.__init__` / `build_optimizer` / logger `save_state` / `from_run_id` run with the build's
`MedTsLLM` registered under `models.model_lookup["medtsllm"]`, everything up to the first forward (tests/golden/boundary_probe.py)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

from helpers import GOLDEN, load_case

PROBE = Path(__file__).resolve().parent / "golden" / "boundary_probe.py"


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_reference_trainer_runs_with_the_plugin_class():
    out = subprocess.run([sys.executable, str(PROBE)], capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("BOUNDARY_PROBE ")]
    assert out.returncode == 0 and line, out.stderr[-3000:]
    r = json.loads(line[-1][len("BOUNDARY_PROBE "):])
    meta, _, _, _ = load_case("gpt2_concat_fc")
    assert r["trainer_class"] == "tasks.forecasting.ForecastTask"                      # the reference's trainer ...
    assert r["model_class"] == "med_ts_llm_amd.models.medtsllm.MedTsLLM"              # ... around the build's model
    assert r["optimizer"] == "AdamW" and r["optimizer_param_names"] == r["trainable"]   # R:tasks/base.py:93: requires_grad params only
    assert r["trainable"] == sorted(n for n, v in meta["param_table"].items() if v["requires_grad"])
    assert r["dtype_device"] == ["torch.float32/cpu"]
    assert "forecasting" in r["supported_tasks"] and set(r["batch_keys"]) == {"x_enc", "y"}
    assert r["prompt0"][0] == "<|endoftext|>" and r["prompt0"][1].startswith("Dataset: synthetic multichannel") and r["prompt0"][-1] == "Time series: "
    assert r["checkpoint_keys"] == ["datetime", "epoch", "model", "run_id", "step"]      # R:loggers/base_logger.py:33-39
    assert r["checkpoint_model_keys"] == meta["state_dict_keys"]                         # no llm.*, no word_embeddings
    assert r["from_run_id_restored"] is True
    assert "ROCm GPU" in r["forward_on_cpu"]                                            # no CPU fallback behind the boundary
    assert r["load_pretrained_keys"] == meta["load_pretrained_keys"]
