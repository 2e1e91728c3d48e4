#!/usr/bin/env python3
"""DESIGN.md section 3's kernel table, generated from the round's recorded measurements (no hand-typed numbers).

  python tools/kernel_table.py r05            # rewrites the block between <!-- kernels:begin --> and <!-- kernels:end -->
  python tools/kernel_table.py r05 --check    # exit 1 if DESIGN.md would change

Inputs (all under profiles/): <round>_bench.json (the bench line of the round's refresh, every figure of the table but the
traffic ratios), <round>_pmc_traffic.txt (HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes beside the algorithmic
bytes, tools/pmc_traffic_refresh.sh), pmc_latest.json (the headline kernel's traffic). Every number in the table can be
found in one of them; the last column names the file.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN = re.compile(r"<!-- kernels:begin[^>]*-->")
END = "<!-- kernels:end -->"


def traffic(path):
  """{kernel substring: (algorithmic bytes, measured bytes, us)} from a pmc_traffic file."""
  alg, meas = {}, {}
  if not os.path.exists(path):
    return {}
  name = None
  for line in open(path):
    if line.startswith("ALG "):
      k, v = line[4:].rstrip("\n").split("\t")
      alg[k] = int(v)
    elif line.startswith("mi355q::"):
      name = line.strip()
    elif name and "total" in line:
      m = re.search(r"mean duration ([\d.]+) us .* total ([\d.]+) MB", line)
      if m:
        meas[name] = (float(m.group(2)) * 1e6, float(m.group(1)))
      name = None
  out = {}
  for k, a in alg.items():
    hit = next((v for n, v in meas.items() if k in n), None)
    if hit:
      out[k] = (a, hit[0], hit[1])
  return out


def f(x, nd=3):
  return "-" if x is None else f"{x:.{nd}f}"


def table(rnd):
  P = os.path.join(ROOT, "profiles")
  b = json.load(open(os.path.join(P, f"{rnd}_bench.json")))
  e, r = b["extras"], b["roofline"]
  tr = traffic(os.path.join(P, f"{rnd}_pmc_traffic.txt"))
  bench = f"`profiles/{rnd}_bench.json`"

  def ratio(key):
    t = tr.get(key)
    return f"{t[1] / t[0]:.2f} (`{rnd}_pmc_traffic.txt`)" if t else "-"
  rows = []
  add = rows.append
  add(("`requant_rows_kernel<8,256,4>` batched, 16 x 4096² int8 per-channel (C2, the headline)", "`mi355q_requant_sym_f32_batched` / `_hostptrs`", "HBM",
       "5.0012 B/elem: 83 906 560 B per buffer", f"{r['launch_ms'] * 1e3:.1f} us per 16", f"{r['frac']:.3f}",
       f"{r['traffic'] / r['alg_bytes_per_launch']:.4f} (`pmc_latest.json`)" if r.get("traffic") else "-", bench + " `roofline`"))
  s = e["single_buffer_launch"]
  add(("the same kernel, one 4096² buffer per launch", "`mi355q_requant_sym_f32`", "HBM", "83 906 560 B", f"{s['ms'] * 1e3:.1f} us", f"{s['hbm_frac']:.3f}", "-", bench + " `extras.single_buffer_launch`"))
  a = e["api_resident"]["c2_int8_channelwise_4096x4096"]
  add(("the same through `get_tensor_quant_params` inside `requant_queue.batching()` (64 resident weights, wall clock)", "public call", "HBM", "83 906 560 B",
       f"{a['us_per_tensor']:.1f} us per tensor", f"{a['hbm_frac']:.3f}", "-", bench + " `extras.api_resident`"))
  c3 = e["c3_blockwise128_int4_packed"]
  add(("`requant_groups_kernel<4,32,1,2>` blockwise-128 int4, packed + f16 scales, 4096 x 11008 (C3 layer)", "`mi355q_requant_sym_f32_batched`", "HBM",
       f"{c3['alg_bytes_per_layer']} B per layer", f"{c3['ms_per_layer'] * 1e3:.1f} us per layer", f"{c3['hbm_frac']:.3f}", ratio("requant_groups_kernel"), bench + " `extras.c3_blockwise128_int4_packed`"))
  a3 = e["api_resident"]["c3_int4_blockwise128_4096x11008"]
  add(("the same through the public call (64 resident weights)", "public call", "HBM", "203 603 968 B", f"{a3['us_per_tensor']:.1f} us per tensor", f"{a3['hbm_frac']:.3f}", "-", bench + " `extras.api_resident`"))
  am = e["c4_act_minmax"]
  add(("`act_minmax_kernel<8>` 128 activations of 4 MiB (C4 statistics)", "`mi355q_act_minmax_f32`", "HBM", "4 B/elem", f"{am['ms_per_launch'] * 1e3:.1f} us per 512 MiB",
       f"{am['roofline']['frac']:.3f}", ratio("act_minmax_kernel"), bench + " `extras.c4_act_minmax`"))
  had = e["hadamard_4096x4096"]
  add(("`fwht_tile_kernel<12>` Hadamard h = 4096 on 4096 rows", "`mi355q_hadamard_rotate_f32`", "HBM", "8 B/elem (read + write)", f"{had['ms'] * 1e3:.1f} us", f"{had['roofline']['frac']:.3f}",
       ratio("fwht_tile_kernel<12"), bench + " `extras.hadamard_4096x4096`"))
  oc = e["octav_clip_4096x4096_int4"]
  add(("`octav_rows_kernel<1,256>` + `octav_tail_kernel` OCTAV clip search 4096², bit-exact (default)", "`mi355q_octav_clip_f32`", "HBM (one read)", "4 B/elem", f"{oc['ms'] * 1e3:.1f} us",
       f"{oc['hbm_frac_of_one_read']:.3f}", ratio("octav_rows_kernel<1, 256>"), bench + " `extras.octav_clip_4096x4096_int4`"))
  ob = e["octav_clip_4096x4096_int4_blockwise128"]
  add(("`octav_unit_lanes_kernel<128>` the same for blocks of 128 (a lane per block), bit-exact", "`mi355q_octav_clip_f32`", "HBM (one read)", "4 B/elem", f"{ob['ms'] * 1e3:.1f} us", f"{ob['hbm_frac_of_one_read']:.3f}",
       ratio("octav_unit_lanes_kernel"), bench + " `extras.octav_clip_4096x4096_int4_blockwise128`"))
  if "octav_clip_4096x4096_int4_blockwise32" in e:
    o32 = e["octav_clip_4096x4096_int4_blockwise32"]
    add(("`octav_unit_lanes_kernel<32>` blocks of 32, bit-exact", "`mi355q_octav_clip_f32`", "HBM (one read)", "4 B/elem", f"{o32['ms'] * 1e3:.1f} us",
         f"{o32['hbm_frac_of_one_read']:.3f}", "-", bench + " `extras.octav_clip_4096x4096_int4_blockwise32`"))
  if "octav_clip_4096x4096_int4_fast" in e:
    of, of2 = e["octav_clip_4096x4096_int4_fast"], e["octav_clip_2048x16384_int4_fast"]
    add(("`octav_fast_kernel<64,16>` one-read OCTAV 4096² (opt-in, T2)", "`mi355q_octav_clip_fast_f32`", "HBM (one read)", "4 B/elem", f"{of['ms'] * 1e3:.1f} us", f"{of['hbm_frac_of_one_read']:.3f}",
         ratio("octav_fast_kernel<64, 16>"), bench + " `extras.octav_clip_4096x4096_int4_fast`"))
    add(("`octav_fast_kernel<1024,16>` the same, 2048 x 16384", "`mi355q_octav_clip_fast_f32`", "HBM (one read)", "4 B/elem", f"{of2['ms'] * 1e3:.1f} us", f"{of2['hbm_frac_of_one_read']:.3f}",
         ratio("octav_fast_kernel<1024, 16>"), bench + " `extras.octav_clip_2048x16384_int4_fast`"))
  # ---- round 6: MSE, OSCAR and the non-fused kernels (bench.py round6_extras)
  if "mse_4096x4096_int4" in e:
    ms_ = e["mse_4096x4096_int4"]
    add(("`mse_scale_balanced_kernel` MSE scale 4096², NumPy pairwise order (bit-exact)", "`mi355q_mse_scale_f32`", "HBM (one read)", "4 B/elem", f"{ms_['scale_kernel_ms'] * 1e3:.1f} us",
         f"{ms_['hbm_frac_of_one_read']:.3f}", ratio("mse_scale_balanced_kernel"), bench + " `extras.mse_4096x4096_int4`"))
    if "scale_and_quantize_kernel_ms" in ms_:
      add(("the same kernel with the quantize behind the sum (one launch: scale + int8)", "`mi355q_mse_requant_f32`", "HBM", "5 B/elem", f"{ms_['scale_and_quantize_kernel_ms'] * 1e3:.1f} us",
           f"{ms_['scale_and_quantize_hbm_frac']:.3f}", "-", bench + " `extras.mse_4096x4096_int4`"))
    add(("MSE `get_tensor_quant_params` on a resident 4096² weight (wall clock)", "public call", "HBM", "5 B/elem (one read, int8 out)" if "scale_and_quantize_kernel_ms" in ms_ else "9 B/elem (two reads, int8 out)", f"{ms_['public_call_ms'] * 1e3:.1f} us",
         f"{ms_['public_call_hbm_frac']:.3f}", "-", bench + " `extras.mse_4096x4096_int4`"))
  for label, key in (("channelwise", "oscar_4096x4096_int4_channelwise"), ("blocks of 128", "oscar_4096x4096_int4_b128")):
    if key in e:
      o = e[key]
      cb = o["clip_bounds"]
      kern = "`clip_prefix_kernel<64,1>` (rows answered from a sorted prefix; full sort + scan for the rest)" if label == "channelwise" else "`sort_tile_kernel` + `clip_scan_kernel<8>`"
      add((f"{kern} OSCAR clip search 4096², {label}, bit-exact", "`mi355q_oscar_clip_bounds_f32`", "HBM (one read)", "4 B/elem", f"{cb['ms'] * 1e3:.1f} us", f"{cb['hbm_frac_of_one_read']:.3f}",
           ratio("clip_prefix_kernel<64, 1>") if label == "channelwise" else "-", bench + f" `extras.{key}.clip_bounds`"))
      add((f"OSCAR `get_tensor_quant_params` 4096², {label}: column energies {e['oscar_col_sumsq_4096x4096']['ms'] * 1e3:.0f} us, 4 x group terms {o['group_terms']['ms'] * 1e3:.0f} us,"
           f" 3 x winner energy {o['winner_energy']['ms'] * 1e3:.0f} us, clip search, quantize {o['quantize']['ms'] * 1e3:.0f} us + the host's O(columns) NumPy between them (wall clock)",
           "public call", "host", "-", f"{o['public_call_ms']:.2f} ms", f"{o['public_call_hbm_frac_of_one_read']:.4f} of one read", "-", bench + f" `extras.{key}`"))
  if "oscar_clip_bounds_2048x16384_channelwise" in e:
    o2 = e["oscar_clip_bounds_2048x16384_channelwise"]
    add(("`clip_prefix_kernel<64,4>` OSCAR clip search 2048 x 16384, channelwise, bit-exact", "`mi355q_oscar_clip_bounds_f32`", "HBM (one read)", "4 B/elem", f"{o2['ms'] * 1e3:.1f} us",
         f"{o2['hbm_frac_of_one_read']:.3f}", "-", bench + " `extras.oscar_clip_bounds_2048x16384_channelwise`"))
  for key, name, entry, alg, tkey in (
      ("minmax_f32_4096x4096_channelwise", "`minmax_runs_kernel` per-channel min / max 4096² (the non-fused route)", "`mi355q_minmax_f32`", "4 B/elem", "minmax_runs_kernel"),
      ("minmax_f32_4096x4096_tensorwise", "the same, TENSORWISE (one channel: 4096 partials, one workgroup combines them)", "`mi355q_minmax_f32`", "4 B/elem", None),
      ("quantize_f32_4096x4096_int8_asymmetric", "`quantize_rows_vec4_kernel` quantize with given scale / zero point, int8 asymmetric", "`mi355q_quantize_f32`", "5 B/elem", "quantize_rows_vec4_kernel"),
      ("dequantize_f32_4096x4096_int8", "`dequantize_rows_vec4_kernel` int8 -> float32", "`mi355q_dequantize_f32`", "5 B/elem", "dequantize_rows_vec4_kernel")):
    if key in e:
      v = e[key]
      add((name, entry, "HBM", alg, f"{v['ms'] * 1e3:.1f} us", f"{v.get('hbm_frac', v.get('hbm_frac_of_one_read')):.3f}", ratio(tkey) if tkey else "-", bench + f" `extras.{key}`"))
  for d in (2048, 16384):
    g = e["c5_gptq"][f"d{d}"]
    h = g["hessian"]
    add((f"`xtx_bf16x3_{'deep_' if d >= 4096 else ''}kernel` GPTQ Hessian d = {d}, {h['tokens']} tokens (exact 3-way bf16 split, default)", "`mi355q_gptq_xtx_f32` / `_accum_f32`", "MFMA bf16",
         "6 n d² flops (triangle)", f"{h['ms']:.2f} ms", f"{h['roofline']['frac']:.3f} ({h['roofline']['achieved']:.0f} TF)", "-", bench + f" `extras.c5_gptq.d{d}.hessian`"))
    hf = h["fast_f16x2"]
    add((f"`xtx_f16x2_kernel` the same, two-way float16 split (opt-in)", "`ops.hessian_product(\"fast\")`", "MFMA f16", "3 n d² flops", f"{hf['ms']:.2f} ms", f"{hf['roofline_frac']:.3f}", "-", bench + f" `...hessian.fast_f16x2`"))
    hi = g["hinv"]
    add((f"damped inverse d = {d} (blocked FP64 Cholesky" + (" with look-ahead; TRTRI + LᵀL on the bf16 split" if d >= 4096 else ", everything FP64") + ")", "`mi355q_gptq_hinv_f64` / `_from_product_f32`",
         "MFMA f64" + (" + bf16" if d >= 4096 else ""), "d³/3 + 2d³/3 flops", f"{hi['ms']:.2f} ms", f"{hi['roofline']['frac']:.3f}", "-", bench + f" `extras.c5_gptq.d{d}.hinv`"))
    ap = g["apply_2048_rows_int4"]
    add((f"`gptq_rows_kernel` + far updates, OBS apply 2048 rows x d = {d}, int4", "`mi355q_gptq_apply_f32`", "MFMA " + ("bf16 (split)" if d >= 4096 else "f32") + " / latency", "2 rows d² flops",
         f"{ap['ms']:.2f} ms", f"{ap['roofline']['frac']:.3f}", ratio("gptq_rows_kernel") if d == 2048 else "-", bench + f" `extras.c5_gptq.d{d}.apply_2048_rows_int4`"))
  sh = b.get("sharded") or {}
  whole = []
  if "c3_32x4096x11008_int4_b128" in sh:
    c = sh["c3_32x4096x11008_int4_b128"]
    whole.append(f"C3 file in -> file out (32 x 4096 x 11008 fp32 -> int4 blockwise-128): {c['seconds']:.3f} s = {c['weight_GBps']:.1f} GB/s of weights (PCIe-bound: `extras.pcie` {e['pcie']['h2d_pinned_GBps']:.0f} GB/s up)")
  if "c4_512_samples_static_wi8_ai8" in sh:
    c = sh["c4_512_samples_static_wi8_ai8"]
    whole.append(f"C4 512 samples x 32 x 4 MiB through `calibrate_sharded` (K samples per launch): {c['seconds'] * 1e3:.1f} ms = {c['activation_GBps']:.0f} GB/s = {c['activation_GBps'] / 8000:.2f} of HBM end to end")
  for key, label in (("c5_gptq", "C5 GPTQ int4, 18 layers, first call of the process"), ("c5_gptq_fast_hessian", "the same with the opt-in f16x2 Hessian product (a later call)"), ("c5_mixed", "C5 mixed (GPTQ + Hadamard/OCTAV on down)")):
    if key in sh:
      c = sh[key]
      extra = ""
      if "second_call_of_the_process" in c:
        extra = f"; second call {c['second_call_of_the_process']['seconds']:.2f} s, busy {c['second_call_of_the_process']['gpu_busy_frac']:.2f}"
      whole.append(f"{label}: {c['seconds']:.2f} s, GPU busy {c['gpu_busy_frac']:.2f}{extra}")
  out = [f"Generated by `tools/kernel_table.py {rnd}` from `profiles/{rnd}_bench.json` (driver-format bench line of the round's refresh: N = 1, {b['steps']} steps),"
         f" `profiles/{rnd}_pmc_traffic.txt` and `profiles/pmc_latest.json`. Fractions are of 8.0 TB/s HBM, 2.5 PFLOP/s bf16 / f16 MFMA, 157.3 TFLOP/s f32 MFMA,"
         " 78.6 TFLOP/s f64 MFMA (`MI355X_MICROARCH.md`); *traffic* = measured HBM bytes / algorithmic bytes per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE passes"
         " with the guide's gfx950 corrections).", "",
         "| Kernel / workload | Entry point | Bound | Algorithmic work | Time | Fraction of peak | Traffic | Source |", "|---|---|---|---|---|---|---|---|"]
  out += ["| " + " | ".join(r_) + " |" for r_ in rows]
  out += ["", "Whole configurations through the public calls (`sharded.*` of the same bench line):", ""] + [f"* {w}" for w in whole]
  out += ["", f"CPU baseline of the same line: {b['cpu_baseline']['value']} {b['cpu_baseline']['unit']} on {b['cpu_baseline']['cores']} core(s)"
          f" (`{b['cpu_baseline']['kind']}`: {b['cpu_baseline'].get('sample', '')}).".replace("  ", " ")]
  return "\n".join(out)


def main():
  args = [a for a in sys.argv[1:] if not a.startswith("--")]
  rnd = args[0] if args else "r06"
  text = table(rnd)
  path = os.path.join(ROOT, "DESIGN.md")
  old = open(path).read()
  m = BEGIN.search(old)
  if not m or END not in old[m.end():]:
    raise SystemExit("DESIGN.md: no kernels:begin / kernels:end markers")
  end = old.index(END, m.end())
  new = old[:m.start()] + f"<!-- kernels:begin (generated by tools/kernel_table.py {rnd}) -->\n" + text + "\n" + old[end:]
  if new != old:
    if "--check" in sys.argv:
      print("DESIGN.md section 3 is out of date")
      sys.exit(1)
    open(path, "w").write(new)


if __name__ == "__main__":
  main()
