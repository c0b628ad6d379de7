"""End-to-end run of the drop-in functions on a RANDOM-INIT model of a real architecture (no weights or datasets are
reachable offline): HF forward passes with the adapter hooks streaming activations into the HIP covariance kernel,
BI scores, keep-ratio allocation, compress_nystrom / compress_qk / compress_vo for every layer, convert_model,
patch_config.  Prints wall-clock per phase.  (dev tool; SURVEY.md 8f-2 "real-model calibration driver")

    python scripts/e2e_random_model.py --arch llama-3-8b --calib_size 64 --batch 16 [--layers 4]
"""
import argparse, os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transformers
from modegpt_amd import engine
from modegpt_amd.adapters.CompressionConfig import CompressionConfig
from modegpt_amd.adapters.model_adapter import ModelAdapter
from modegpt_amd.calibration import load_calibs
from modegpt_amd.compression.compress_mlp import compress_nystrom
from modegpt_amd.compression.compress_qk import compress_qk
from modegpt_amd.compression.compress_vo import compress_vo
from modegpt_amd.compression_utils import allocate_global_sparsity

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="llama-3-8b", choices=["llama-3-8b", "llama-2-7b", "qwen3-14b", "tiny"])
ap.add_argument("--calib_size", type=int, default=64)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--layers", type=int, default=0, help="truncate the model to this many layers (0 = all)")
ap.add_argument("--ratio", type=float, default=0.3)
a = ap.parse_args()
sh = engine.SHAPES[a.arch]
L = a.layers or sh["n_layers"]
kw = dict(hidden_size=sh["d"], intermediate_size=sh["d_ff"], num_hidden_layers=L, num_attention_heads=sh["n_heads"],
          num_key_value_heads=sh["n_kv_heads"], head_dim=sh["head_dim"], vocab_size=32000, max_position_embeddings=2048)
dev = torch.device("cuda:0")
torch.manual_seed(0)
t0 = time.time()
with torch.device(dev):
    torch.set_default_dtype(torch.bfloat16)
    if sh["arch"] == "qwen3":
        model = transformers.Qwen3ForCausalLM(transformers.Qwen3Config(**kw))
    else:
        model = transformers.LlamaForCausalLM(transformers.LlamaConfig(**kw))
    torch.set_default_dtype(torch.float32)
model.eval()
print(f"built {a.arch} x{L} layers, {sum(p.numel() for p in model.parameters())/1e9:.2f} B params in {time.time()-t0:.1f} s")
tmp = tempfile.mkdtemp(prefix="mdg_e2e_")
ad = ModelAdapter.from_model(model, None)
ad.config = CompressionConfig(temp_storage_dir=os.path.join(tmp, "layers"), dataset="synthetic", calib_size=a.calib_size,
                              calibs_batch_size=a.batch, compression_ratio=a.ratio, order="mlp,qk,vo", **engine.RECIPE_RIDGES)
torch.cuda.synchronize(); t0 = time.time()
cov_mlp, cov_q, cov_k, cov_x, bi = load_calibs(ad, a.calib_size, a.batch, dataset="synthetic", target_layers=[])
torch.cuda.synchronize(); t_cal = time.time() - t0
print(f"calibration ({a.calib_size} x 2048 tokens, all {L} layers hooked): {t_cal:.1f} s; bi[:4] = {[round(b, 4) for b in bi[:4]]}")
keep = allocate_global_sparsity(bi, a.ratio, smoothing=ad.config.sparsity_smoothing, max_sparsity=ad.config.max_sparsity, adapter=ad)
layers = list(range(L))
torch.cuda.synchronize(); t0 = time.time(); compress_nystrom(ad, cov_mlp, keep, layers); torch.cuda.synchronize(); t_mlp = time.time() - t0
t0 = time.time(); masks = compress_qk(ad, (cov_q, cov_k), keep, target_layers=layers); torch.cuda.synchronize(); t_qk = time.time() - t0
t0 = time.time(); compress_vo(ad, cov_x, keep, target_layers=layers); torch.cuda.synchronize(); t_vo = time.time() - t0
print(f"compress (incl. torch.save of the artefacts): mlp {t_mlp:.1f} s, qk {t_qk:.1f} s, vo {t_vo:.1f} s")
del cov_mlp, cov_q, cov_k, cov_x
t0 = time.time(); ad.convert_model(saved_layers_dir=ad.config.temp_storage_dir); ad.patch_config(); t_cv = time.time() - t0
cfg = model.config
print(f"convert_model + patch_config: {t_cv:.1f} s; gate_ranks[:4] {cfg.gate_ranks[:4]} q_ranks[:2] {cfg.q_ranks[:2]} v_ranks[:2] {cfg.v_ranks[:2]} mask {tuple(masks[0].shape)}")
from modegpt_amd.patchers import install_compressed_attention
from modegpt_amd.eval import compute_perplexity
install_compressed_attention(ad, masks)
t0 = time.time(); ppl = compute_perplexity(model, None, bs=4, dataset="synthetic", adapter=ad); t_ppl = time.time() - t0
print(f"compressed model, in-process compressed attention: synthetic-token perplexity {ppl:.1f} (random weights; vocab 32000) in {t_ppl:.1f} s")
print(f"TOTAL {t_cal + t_mlp + t_qk + t_vo:.1f} s for {L} layers = {L / (t_cal + t_mlp + t_qk + t_vo):.3f} layers/s (model forward and artefact IO included)")
shutil.rmtree(tmp, ignore_errors=True)
from modegpt_amd import ops as _ops
print(f"covariance routes of the large statistics (mode {_ops.COV_MODE}; counted on the device): {getattr(ad, 'cov_routes', None)}")
