"""THIS container only (needs /root/reference; not part of any test or bench): time the actual reference (stub-imported) next to the oracle port, same model/shapes (SURVEY 8d iii)."""
import sys, time, tempfile, os, json
from pathlib import Path
sys.path.insert(0, "/root/repo/tests/golden"); sys.path.insert(0, "/root/repo")
import torch, numpy as np
import make_golden as MG
torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
with tempfile.TemporaryDirectory() as tmp:
    MG.setup_imports(tmp)
    from transformers import GPT2Config, GPT2Model
    d = str(Path(tmp) / "gpt2s"); os.makedirs(d)
    torch.manual_seed(0)
    cfg = GPT2Config(resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    GPT2Model(cfg).save_pretrained(d)
    MG.make_tokenizer(d, vocab=384)
    import models as ref_models
    from utils import dict_to_object
    B, L, C, pred = 4, 1024, 12, 96
    c = MG.base_config(d, "forecasting", L, pred, "concat", "linear", MG.PROMPT_CONST, d_model=32, d_ff=128, H=8, num_tokens=1024)
    model = ref_models.model_lookup["medtsllm"](dict_to_object(c), MG.DS(C)).to("cpu", torch.float32)
    model.train()
    x = torch.randn(B, L, C); y = torch.randn(B, pred, C)
    def step():
        loss = torch.nn.functional.mse_loss(model({"x_enc": x}), y); loss.backward(); model.zero_grad()
    step()
    t0 = time.perf_counter(); n = 3
    for _ in range(n): step()
    dt = (time.perf_counter() - t0) / n
    print(f"reference: {dt:.2f} s/step -> {B / dt:.2f} samples/s at {torch.get_num_threads()} threads")
    # oracle on the same weights
    from oracle import medtsllm_oracle as O
    sd = {k: v.detach() for k, v in model.llm.state_dict().items()}
    p = {n_: t.detach().clone().requires_grad_(True) for n_, t in model.named_parameters() if t.requires_grad}
    bcfg = json.loads(open(os.path.join(d, "config.json")).read())
    prompts = model.build_prompt({"x_enc": x})
    tok = [[model.tokenizer(s, padding=False, truncation=False).input_ids for s in ps] for ps in prompts]
    m = {"task": "forecasting", "pred_len": pred, "patch_len": 16, "stride": 8, "n_heads": 8, "d_ff": 128, "covariate_mode": "concat",
         "embedding_downsample_mode": "linear", "n_outputs_per_step": C, "n_classes": 0}
    def ostep():
        out = O.medtsllm_forward(x, p, sd, bcfg, m, token_ids=tok, pad_token_id=model.tokenizer.pad_token_id, training=True)
        torch.nn.functional.mse_loss(out, y).backward()
        for t in p.values(): t.grad = None
    ostep()
    t0 = time.perf_counter()
    for _ in range(n): ostep()
    dto = (time.perf_counter() - t0) / n
    print(f"oracle   : {dto:.2f} s/step -> {B / dto:.2f} samples/s   (oracle/reference time ratio {dto / dt:.2f})")
