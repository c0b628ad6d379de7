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
ap.add_argument("--price", action="store_true",
                help="price the run (VERDICT r2 item 5): a forward-only pass (hooks off), the hooks' kernel time by HIP events around "
                     "every covariance enqueue, BI time, artefact IO, and the per-rank projection at 8 GPUs")
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
price = {}
if a.price:
    from modegpt_amd import ops as _ops, calibration as _cal
    from modegpt_amd.eval import load_calibration_texts
    ad.calibs = load_calibration_texts(calib_size=a.calib_size, model=model, tokenizer=None, batch_size=a.batch, dataset="synthetic")
    with torch.no_grad():
        model(ad.calibs[0])                                            # warm-up (allocator, kernels)
        torch.cuda.synchronize(); t0 = time.time()
        for batch in ad.calibs:
            model(batch)
        torch.cuda.synchronize(); price["forward_only_s"] = time.time() - t0
        torch.cuda.synchronize(); t0 = time.time()
        for batch in ad.calibs:
            out = model(batch, output_hidden_states=True); del out
        torch.cuda.synchronize(); price["forward_with_hidden_states_s"] = time.time() - t0
    ev = []
    real_multi = _ops.cov_accum_multi
    def timed_multi(items, mode=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); real_multi(items, mode); e1.record(); ev.append((e0, e1))
    _ops.cov_accum_multi = timed_multi
    bi_ev = []
    real_add = _cal.BlockInfluence.add_batch
    def timed_add(self, hs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.time(); e0.record(); real_add(self, hs); e1.record(); bi_ev.append((e0, e1, time.time() - t0))
    _cal.BlockInfluence.add_batch = timed_add
    io = [0.0]
    real_save = type(ad).save_layer
    def timed_save(self, *args, **kw):
        torch.cuda.synchronize(); t0 = time.time(); real_save(self, *args, **kw); io[0] += time.time() - t0
    type(ad).save_layer = timed_save
torch.cuda.synchronize(); t0 = time.time()
cov_mlp, cov_q, cov_k, cov_x, bi = load_calibs(ad, a.calib_size, a.batch, dataset="synthetic", target_layers=[])
torch.cuda.synchronize(); t_cal = time.time() - t0
if a.price:
    price["hook_kernels_s"] = sum(e0.elapsed_time(e1) for e0, e1 in ev) * 1e-3
    price["hook_enqueues"] = len(ev)
    price["bi_kernels_s"] = sum(e0.elapsed_time(e1) for e0, e1, _ in bi_ev) * 1e-3
    price["bi_host_s"] = sum(h for _, _, h in bi_ev)
print(f"calibration ({a.calib_size} x 2048 tokens, all {L} layers hooked): {t_cal:.1f} s; bi[:4] = {[round(b, 4) for b in bi[:4]]}")
keep = allocate_global_sparsity(bi, a.ratio, smoothing=ad.config.sparsity_smoothing, max_sparsity=ad.config.max_sparsity, adapter=ad)
layers = list(range(L))
from modegpt_amd.compression import _window


def compress_all():
    torch.cuda.synchronize(); t0 = time.time(); compress_nystrom(ad, cov_mlp, keep, layers); torch.cuda.synchronize(); t1 = time.time()
    masks = compress_qk(ad, (cov_q, cov_k), keep, target_layers=layers); torch.cuda.synchronize(); t2 = time.time()
    compress_vo(ad, cov_x, keep, target_layers=layers); torch.cuda.synchronize(); t3 = time.time()
    ad.flush_artifacts(); t4 = time.time()
    return masks, t1 - t0, t2 - t1, t3 - t2, t4 - t3


if a.price:
    # the reference's order first (one layer after the other, torch.save before the next starts), then what run_modegpt does
    width, _window.CHAIN_WIDTH = _window.CHAIN_WIDTH, 1
    masks, t_mlp, t_qk, t_vo, _ = compress_all()
    _window.CHAIN_WIDTH = width
    type(ad).save_layer = real_save
    price["reference_order"] = (t_mlp, t_qk, t_vo, io[0])
    print(f"compress, reference order (one layer at a time, torch.save in line): mlp {t_mlp:.1f} s, qk {t_qk:.1f} s, vo {t_vo:.1f} s")
ad.async_artifacts(True)
masks, t_mlp, t_qk, t_vo, t_flush = compress_all()
print(f"compress ({_window.CHAIN_WIDTH} layers' chains in flight, artefacts through the background writer): mlp {t_mlp:.1f} s, qk {t_qk:.1f} s, "
      f"vo {t_vo:.1f} s, waiting for the writer at the end {t_flush:.2f} s")
t_vo += t_flush
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
if a.price:
    fw, hk, bik = price["forward_with_hidden_states_s"], price["hook_kernels_s"], price["bi_kernels_s"]
    r_mlp, r_qk, r_vo, r_io = price["reference_order"]
    dec = r_mlp + r_qk + r_vo - r_io
    print(f"PRICE calibration {t_cal:.1f} s = forward {fw:.1f} (without output_hidden_states: {price['forward_only_s']:.1f}) + hooks' covariance "
          f"kernels {hk:.1f} ({price['hook_enqueues']} enqueues) + BI kernels {bik:.2f} (host side of BI incl. its syncs {price['bi_host_s']:.2f}) "
          f"+ rest (host stalls, allocator, finalize) {t_cal - fw - hk - bik:.1f}")
    print(f"PRICE compress, reference order {r_mlp + r_qk + r_vo:.1f} s = kernels + host {dec:.1f} + artefact IO (torch.save after a synchronize) {r_io:.1f};  "
          f"as run_modegpt runs it {t_mlp + t_qk + t_vo:.1f} s")
    dec = t_mlp + t_qk + t_vo
    io[0] = 0.0
    G = 8
    print(f"PROJECTION at {G} GPUs, layer-sharded (every rank: all samples through the whole model, hooks on its {L // G} layers): "
          f"forward {fw:.1f} + cov {hk / G:.1f} + BI {bik:.2f} + dec {dec / G:.1f} + IO {io[0] / G:.1f} = "
          f"{fw + hk / G + bik + dec / G + io[0] / G:.1f} s per rank  -> speed-up {(t_cal + t_mlp + t_qk + t_vo) / (fw + hk / G + bik + dec / G + io[0] / G):.2f}x; "
          f"with the forward truncated after a rank's last owned layer (mean (g + 1) / G of it over ranks, the LAST rank the full one): "
          f"slowest rank unchanged; with BI from one shared bf16 pre-pass and hooks fed by a layer-pipelined forward: see DESIGN.md section 6")
shutil.rmtree(tmp, ignore_errors=True)
from modegpt_amd import ops as _ops
print(f"covariance routes of the large statistics (mode {_ops.COV_MODE}; counted on the device): {getattr(ad, 'cov_routes', None)}")
