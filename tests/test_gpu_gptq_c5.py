"""GPTQ parity at C5's real shapes (Gemma-2B: d = 2048 for q/k/v/o/gate/up inputs, d = 16384 for
down_proj [2048, 16384]; ref algorithms/uniform_quantize/gptq.py:55-128, 131-216).

What is compared with what:
  * Hessian (a10): against the oracle's `x.T.dot(x)` on column subsets (H[i][j] depends on columns
    i and j only) and against an FP64 product computed on the device (torch, checker only);
  * inverse (a11): against the exact FP64 inverse of the damped matrix (torch.linalg on the
    device, checker only) and through the residual |Hinv . Hdamped - I|, with the error level of the
    d = 2048 / 4608 cases recorded beside it;
  * OBS apply (a12): against the oracle on row slices (rows are independent, ref :131-216), fed the
    GPU's Hinv so that ONLY the apply step is compared; the K = 256 lazy updates, the 16-lane block
    kernel (>= 8192 rows) and gemm_fast_big_kernel all run at these sizes.
Every tolerance-class comparison reports its observed rate (tests/parity_rates.py)."""
import numpy as np
import pytest

import parity_rates
from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu

D_BIG = 16384


@pytest.fixture(scope="module")
def m():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as g
  g.build()
  import types
  from mi355q import ops, qtyping, runtime
  from mi355q.algorithms.uniform_quantize import gptq
  return types.SimpleNamespace(ops=ops, q=qtyping, gptq=gptq, torch=torch, rt=runtime)


def _activations(torch, samples, tokens, d, seed):
  """[samples, tokens, d] float32 on the device: unit normal, sixteen loud channels, one dead."""
  gen = torch.Generator(device="cuda").manual_seed(seed)
  x = torch.randn((samples, tokens, d), generator=gen, device="cuda", dtype=torch.float32)
  x[..., 5:21] *= 6.0
  x[..., 77] = 0.0          # dead channel: zero diagonal entry -> 1 (ref :113-114)
  return x


def _xtx(m, x2, alpha, kernel):
  """The Hessian product under one of the two split kernels of the build: "bf16x3" (the default: the exact
  three-way split) or "f16x2" (22-23 of the 24 mantissa bits; MI355Q_XTX_F16X2=1, read per call).
  A precision change in a10 then shows as a RATE change in the a12 comparisons that are run on both."""
  import os
  assert kernel in ("f16x2", "bf16x3")
  if kernel == "f16x2":
    os.environ["MI355Q_XTX_F16X2"] = "1"
  try:
    h = m.ops.gptq_xtx(x2, alpha)
    m.torch.cuda.synchronize()
  finally:
    os.environ.pop("MI355Q_XTX_F16X2", None)
  return h


@pytest.fixture(scope="module")
def big(m):
  """d = 16384: activations of 64 samples x 512 tokens (twice d: the Hessian has full rank, as with
  BASELINE's 128 x 512; with d tokens or fewer only the damping makes it invertible and every
  rounding difference is amplified a hundredfold), their Hessian and its damped inverse, all
  resident in HBM and shared by the tests below."""
  torch = m.torch
  x = _activations(torch, 64, 512, D_BIG, 5000)
  h = m.ops.gptq_xtx(x.reshape(-1, D_BIG), 2.0 / 64)
  hinv, info = m.ops.gptq_hinv(h, 0.01)
  assert int(info.item()) == 0
  torch.cuda.synchronize()
  return {"x": x, "h": h, "hinv": hinv}


def test_hessian_d16384_against_oracle_columns_and_fp64(m, big):
  torch = m.torch
  x, h = big["x"], big["h"]
  assert h.dtype == torch.float64 and tuple(h.shape) == (D_BIG, D_BIG)
  assert torch.equal(h, h.T)
  cols = np.r_[0:128, 8000:8128, D_BIG - 128:D_BIG]
  sub = x[..., torch.from_numpy(cols).cuda()].cpu().numpy()          # [64, 512, 384]
  ref = O.gptq_hessian(sub)                                          # NumPy sgemm, (2/64) x^T x
  got = h[torch.from_numpy(cols).cuda()][:, torch.from_numpy(cols).cuda()].cpu().numpy()
  parity_rates.check_rel("hessian d=16384 32768 tokens vs oracle (384 columns)", got, ref, 2e-6)
  x2 = x.reshape(-1, D_BIG)
  exact = torch.zeros((D_BIG, D_BIG), dtype=torch.float64, device="cuda")
  for k0 in range(0, x2.shape[0], 4096):                             # FP64 checker in K slabs
    xs = x2[k0:k0 + 4096].double()
    exact.addmm_(xs.T, xs)
  exact *= 2.0 / 64
  err = float((h - exact).abs().max() / exact.abs().max())
  parity_rates.note("hessian d=16384 32768 tokens vs FP64 product", "max_rel_error", err, 2e-6)   # bf16 split, two-level FP32 sums: observed 2e-7 (FP32 MFMA: 3.9e-6)


def _damped(torch, h):
  dg = torch.diagonal(h).clone()
  dg = torch.where(dg == 0, torch.ones_like(dg), dg)
  out = h.clone()
  out.diagonal().copy_(dg + 0.01 * dg.mean())
  return out


@pytest.mark.parametrize("d,tokens", [(2048, 8192), (4608, 9216)])
def test_hessian_inverse_error_level_small_shapes(m, d, tokens):
  """The error level d = 16384 is held against (same generator, same checker)."""
  torch = m.torch
  x = _activations(torch, 8, tokens // 8, d, 5100 + d)
  h = m.ops.gptq_xtx(x.reshape(-1, d), 2.0 / 8)
  hinv, info = m.ops.gptq_hinv(h, 0.01)
  assert int(info.item()) == 0
  exact = torch.linalg.inv(_damped(torch, h))
  err = float((hinv.double() - exact).abs().max() / exact.abs().max())
  # d = 2048: FP64 MFMA throughout (3.3e-8); d >= 4096: the single-precision steps of the reference
  # (strtri, the L^-T L^-1 einsum: ref gptq.py:121-128) run on the bf16 split with float32-class accuracy --
  # the bound is the level the reference's own float32 steps reach (6.8e-7)
  parity_rates.note(f"hinv d={d} vs exact FP64 inverse", "max_rel_error", err, 3e-7 if d < 4096 else 7e-7)


def test_hessian_inverse_d16384(m, big):
  torch = m.torch
  h, hinv = big["h"], big["hinv"]
  assert hinv.dtype == torch.float32 and torch.equal(hinv, hinv.T)
  damped = _damped(torch, h)
  # residual: independent of any other inversion routine
  resid = hinv.double() @ damped
  resid.diagonal().sub_(1.0)
  r = float(resid.abs().max())
  del resid
  parity_rates.note("hinv d=16384 residual max|Hinv.Hd - I|", "max_abs_residual", r, 2e-5)        # float32-class triangular inverse and product (FP64 throughout: 5e-7)
  exact = torch.linalg.inv(damped)
  err = float((hinv.double() - exact).abs().max() / exact.abs().max())
  parity_rates.note("hinv d=16384 vs exact FP64 inverse", "max_rel_error", err, 7e-7)   # observed 1.2e-7 on the bf16 split (FP64 throughout: 2.7e-8); the reference's float32 steps: 6.8e-7
  again, _ = m.ops.gptq_hinv(h, 0.01)
  assert torch.equal(hinv, again)          # same launches in the same order: bit-identical


def _channelwise_scale(torch, w, bits):
  qmax = float((1 << (bits - 1)) - 1)
  return (torch.clamp(w.abs().amax(dim=1), min=1e-9) / qmax).contiguous()


def _oracle_rows(w_rows, scale_rows, hinv_host, bits, gran="CHANNELWISE", block=0):
  scale = scale_rows.reshape(w_rows.shape[0], -1).astype(np.float32)
  zp = np.zeros(scale.shape, np.int8)
  return O.gptq_apply(w_rows, scale, zp, bits, True, None, gran, block_size=block, hinv=hinv_host)


def _oracle_rows_split_matmul(w_rows, scale_rows, hinv_host, bits):
  """O.gptq_apply (channelwise, symmetric) with ONE difference: the update of the columns behind a
  64-column block (ref gptq.py:213-214, `W[:, rest] -= err @ Hinv[blk, rest]`) is formed as two
  K = 32 products added together -- the same float32 products in another addition order."""
  fw = np.array(w_rows, copy=True)
  scale = scale_rows.reshape(-1, 1).astype(np.float32)
  zp = np.zeros(scale.shape, np.int8)
  qw = np.zeros(fw.shape, np.int8)
  d = hinv_host.shape[0]
  for b0 in range(0, d, 64):
    b1 = min(b0 + 64, d)
    wb = fw[:, b0:b1]
    eb = np.zeros_like(wb)
    for i in range(b1 - b0):
      c = b0 + i
      col = wb[:, i]
      qc = O.uniform_quantize(np.expand_dims(col, -1), scale, zp, bits, True, quantized_dim=0).reshape(-1, 1)
      dq = O.uniform_dequantize(qc, scale, zp, quantized_dim=0).reshape(-1)
      qw[:, c] = qc.reshape(-1)
      np.subtract(col, dq, out=eb[:, i])
      eb[:, i] /= hinv_host[c, c]
      if i < b1 - b0 - 1:
        wb[:, i + 1:] -= np.outer(eb[:, i], hinv_host[c, c + 1:b1])
    half = (b1 - b0) // 2
    fw[:, b1:] -= np.matmul(eb[:, :half], hinv_host[b0:b0 + half, b1:]) + np.matmul(eb[:, half:], hinv_host[b0 + half:b1, b1:])
  return qw


@pytest.mark.parametrize("kernel", ["bf16x3", "f16x2"])
def test_apply_down_proj_2048x16384_int8_rows_against_oracle(m, big, kernel):
  """int8 at the down_proj shape (scales 18 x finer than int4's): GPU vs oracle with the same
  inverse, beside the oracle's own reproducibility under another block-update summation order
  (three rates: parity_rates.check_with_floor). Run on the inverse of BOTH Hessian kernels' products:
  the instance differs, the apply step under test does not."""
  torch = m.torch
  if kernel == "bf16x3":
    hinv = big["hinv"]
  else:
    hinv, info = m.ops.gptq_hinv(_xtx(m, big["x"].reshape(-1, D_BIG), 2.0 / 64, kernel), 0.01)
    assert int(info.item()) == 0
  gen = torch.Generator(device="cuda").manual_seed(5210)
  w = torch.randn((2048, D_BIG), generator=gen, device="cuda") * 0.02
  scale = _channelwise_scale(torch, w, 8)
  q = m.ops.gptq_apply(w, hinv, scale, None, 1, 0, 8, True, False, 8)
  hinv_host = hinv.cpu().numpy()
  del hinv
  rows = np.r_[0:8, 2040:2048]
  idx = torch.from_numpy(rows).cuda()
  wr, sr = w[idx].cpu().numpy(), scale[idx].cpu().numpy()
  ref = _oracle_rows(wr, sr, hinv_host, 8)
  ref_b = _oracle_rows_split_matmul(wr, sr, hinv_host, 8)
  name = f"gptq apply [2048,16384] int8 channelwise, 16 rows vs oracle (same Hinv; Hessian by {kernel})"
  if kernel == "bf16x3":      # the default product: its own bound (recorded 0), the floor only recorded beside it
    parity_rates.check_default_path(name, q[idx].cpu().numpy(), ref, ref_b)
  else:                       # the opt-in fast product: floor-based, cap = 2 x the 1.66e-3 of profiles/r03_parity_rates.txt
    parity_rates.check_with_floor(name, q[idx].cpu().numpy(), ref, ref_b, cap=3.4e-3, k=2.0)
    with pytest.raises(AssertionError, match="default-path bound"):       # NEGATIVE CONTROL: the default gate sees this precision
      parity_rates.check_default_path("NEGATIVE CONTROL (must fail): f16x2 Hessian through the default-path gate, apply [2048,16384] int8",
                                      q[idx].cpu().numpy(), ref, ref_b, negative_control=True)


def test_apply_down_proj_2048x16384_rows_against_oracle(m, big):
  """W [2048, 16384] (down_proj): 256 column blocks, 64 K = 256 lazy group updates over a trailing
  matrix of up to 2048 x 16128; oracle on 24 rows with the same Hinv."""
  torch = m.torch
  gen = torch.Generator(device="cuda").manual_seed(5200)
  w = torch.randn((2048, D_BIG), generator=gen, device="cuda") * 0.02
  scale = _channelwise_scale(torch, w, 4)
  q = m.ops.gptq_apply(w, big["hinv"], scale, None, 1, 0, 4, False, False, 8)
  hinv_host = big["hinv"].cpu().numpy()
  rows = np.r_[0:8, 1000:1008, 2040:2048]
  idx = torch.from_numpy(rows).cuda()
  ref = _oracle_rows(w[idx].cpu().numpy(), scale[idx].cpu().numpy(), hinv_host, 4)
  parity_rates.check("gptq apply [2048,16384] int4 channelwise, 24 rows vs oracle (same Hinv)",
                     q[idx].cpu().numpy(), ref, parity_rates.T2)       # observed 0
  qh = q.cpu().numpy()
  assert qh.min() >= -8 and qh.max() <= 7 and (qh != 0).mean() > 0.5


@pytest.mark.parametrize("kernel", ["bf16x3", "f16x2"])
def test_apply_gate_proj_16384x2048_rows_against_oracle(m, kernel):
  """W [16384, 2048] (gate / up): >= 8192 rows take the 16-lanes-per-row block kernel. On the
  inverse of both Hessian kernels' products (see _xtx)."""
  torch = m.torch
  d = 2048
  x = _activations(torch, 16, 512, d, 5300)
  h = _xtx(m, x.reshape(-1, d), 2.0 / 16, kernel)
  hinv, info = m.ops.gptq_hinv(h, 0.01)
  assert int(info.item()) == 0
  gen = torch.Generator(device="cuda").manual_seed(5301)
  w = torch.randn((16384, d), generator=gen, device="cuda") * 0.02
  hinv_host = hinv.cpu().numpy()
  rows = np.r_[0:48, 8192:8240, 16384 - 32:16384]
  idx = torch.from_numpy(rows).cuda()
  for bits, gran, block in ((4, "CHANNELWISE", 0), (4, "BLOCKWISE_32", 32), (8, "CHANNELWISE", 0)):
    if block:
      scale_h = O.min_max_quant_params(w[idx].cpu().numpy(), bits, True, gran)["scale"]
      full = O.blockwise_scale_round(
          (torch.clamp(w.reshape(16384, d // block, block).abs().amax(dim=2), min=1e-9)
           / float((1 << (bits - 1)) - 1)).cpu().numpy())
      assert np.array_equal(full[rows], scale_h)
      scale = torch.from_numpy(np.ascontiguousarray(full.reshape(-1))).cuda()
      q = m.ops.gptq_apply(w, hinv, scale, None, 2, block, bits, False, False, 8)
      ref = _oracle_rows(w[idx].cpu().numpy(), scale_h, hinv_host, bits, gran, block)
    else:
      scale = _channelwise_scale(torch, w, bits)
      q = m.ops.gptq_apply(w, hinv, scale, None, 1, 0, bits, bits >= 8, False, 8)
      ref = _oracle_rows(w[idx].cpu().numpy(), scale[idx].cpu().numpy(), hinv_host, bits)
    if bits == 8:
      # int8: the reference's own reproducibility first -- the same oracle with the update behind a
      # block summed in two halves (what a BLAS with another K blocking does with gptq.py:213-214's
      # matmul): the rate at which THAT flips integers is the floor for any implementation
      ref_b = _oracle_rows_split_matmul(w[idx].cpu().numpy(), scale[idx].cpu().numpy(), hinv_host, bits)
      name = f"gptq apply [16384,2048] int{bits} {gran}, 128 rows vs oracle (same Hinv; Hessian by {kernel})"
      if kernel == "bf16x3":    # the default product: its own bound (recorded 0)
        parity_rates.check_default_path(name, q[idx].cpu().numpy(), ref, ref_b)
      else:                     # opt-in fast product; cap: 2 x the 2.4e-4 recorded in profiles/r03_parity_rates.txt
        parity_rates.check_with_floor(name, q[idx].cpu().numpy(), ref, ref_b, cap=5e-4, k=2.0)
    else:
      parity_rates.check(f"gptq apply [16384,2048] int{bits} {gran}, 128 rows vs oracle (same Hinv; Hessian by {kernel})",
                         q[idx].cpu().numpy(), ref, parity_rates.T2)


def test_apply_split_update_with_an_odd_number_of_column_tiles(m):
  """d = 4224 = 33 tiles of 128 columns: the update behind a group runs on the bf16 matrix cores
  (d >= 4096) two column tiles per workgroup, and behind every other group the last workgroup of
  a row has a single tile left. 16 rows against the oracle with the same Hinv."""
  torch = m.torch
  d = 4224
  x = _activations(torch, 16, 640, d, 5500)
  hinv, info = m.ops.gptq_hinv(m.ops.gptq_xtx(x.reshape(-1, d), 2.0 / 16), 0.01)
  assert int(info.item()) == 0
  gen = torch.Generator(device="cuda").manual_seed(5501)
  w = torch.randn((128, d), generator=gen, device="cuda") * 0.02
  scale = _channelwise_scale(torch, w, 4)
  q = m.ops.gptq_apply(w, hinv, scale, None, 1, 0, 4, False, False, 8)
  rows = np.r_[0:8, 120:128]
  idx = torch.from_numpy(rows).cuda()
  ref = _oracle_rows(w[idx].cpu().numpy(), scale[idx].cpu().numpy(), hinv.cpu().numpy(), 4)
  parity_rates.check("gptq apply [128,4224] int4 channelwise, 16 rows vs oracle (same Hinv)",
                     q[idx].cpu().numpy(), ref, parity_rates.T2)


def test_down_proj_through_get_tensor_quant_params(m, big):
  """The public entry point at the C5 shape: Hessian handed over as the HBM resident the
  calibrator produces, result rows against the oracle fed the same inverse."""
  torch = m.torch
  q_ = m.q
  cfg = q_.TensorQuantizationConfig(num_bits=4, symmetric=True, granularity=q_.QuantGranularity.CHANNELWISE)
  info = q_.OpInfo(op=q_.OperatorT(), op_name=q_.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                   op_quant_config=q_.OpQuantizationConfig(weight_tensor_config=cfg))
  w = (np.random.default_rng(5400).standard_normal((256, D_BIG), dtype=np.float32) * np.float32(0.02))
  hess = m.rt.HbmArray(big["h"])
  p = m.gptq.get_tensor_quant_params(info, cfg, w, {"activation_tensor_qsv": {"hessian": hess, "num_samples": 64}})
  ref_scale = O.min_max_quant_params(w, 4, True, "CHANNELWISE")["scale"]
  assert np.array_equal(p.scale, ref_scale)
  hinv_host = hess.cache[("hinv", 0.01)][0].cpu().numpy()
  assert np.array_equal(hinv_host, big["hinv"].cpu().numpy())       # one inverse per Hessian, deterministic
  rows = np.r_[0:8, 248:256]
  ref = _oracle_rows(w[rows], ref_scale[rows], hinv_host, 4)
  parity_rates.check("gptq.get_tensor_quant_params [256,16384] int4, 16 rows vs oracle (same Hinv)",
                     np.asarray(p.quantized_data)[rows], ref, parity_rates.T2)


def test_full_chain_with_llm_like_activations_d4096(m):
  """Activations shaped like a decoder's, not like white noise: per-token scales spread over a decade
  (log-normal), eight massive channels 60 x the rest, a third of the entries gated to (almost) zero, a non-zero
  mean per channel -- at d = 4096, on the default Hessian product (the exact three-way bfloat16 split, 128 x 256 tiles). The whole chain -- Hessian, damped inverse,
  OBS apply through get_tensor_quant_params -- against the oracle's own chain (float32 x.T.dot(x), FP64
  Cholesky, single-precision triangular inverse and product: ref gptq.py:100-216), with the oracle's own
  reproducibility (its Hessian summed in two halves) measured beside it."""
  torch, q_ = m.torch, m.q
  d, tokens, rows = 4096, 8192, 48
  gen = torch.Generator(device="cuda").manual_seed(4242)
  x = torch.randn((tokens, d), generator=gen, device="cuda")
  x = x * torch.exp(0.8 * torch.randn((tokens, 1), generator=gen, device="cuda"))          # per-token scale
  x = x + 0.3 * torch.randn((1, d), generator=gen, device="cuda")                           # per-channel mean
  x[:, torch.randint(0, d, (8,), generator=gen, device="cuda")] *= 60.0                     # massive channels
  x = torch.where(torch.rand((tokens, d), generator=gen, device="cuda") < 0.33, x * 1e-4, x).contiguous()
  xs = x.reshape(16, tokens // 16, d)                                                        # 16 samples
  h = m.ops.gptq_xtx(x, 2.0 / 16)
  ref64 = (x.double().T @ x.double()) * (2.0 / 16)
  mag = (x.double().abs().T @ x.double().abs()) * (2.0 / 16)
  parity_rates.note("hessian f16 split, LLM-like activations d=4096 vs FP64 product (per-entry scale)", "max_rel_error",
                    float(((h - ref64).abs() / mag.clamp_min(1e-300)).max()), 1e-6)
  del ref64, mag
  xh = xs.cpu().numpy()
  hess = O.gptq_hessian(xh)
  x2 = xh.reshape(-1, d)
  half = x2.shape[0] // 2
  hess_b = (2.0 / np.array(16)) * (x2[:half].T.dot(x2[:half]) + x2[half:].T.dot(x2[half:]))
  del xh, x2
  parity_rates.check_rel("LLM-like activations: Hessian d=4096 vs oracle x.T.dot(x)", h.cpu().numpy(), hess, 2e-6)
  w = np.random.default_rng(4243).standard_normal((rows, d), dtype=np.float32) * np.float32(0.02)
  cfg = q_.TensorQuantizationConfig(num_bits=4, symmetric=True, granularity=q_.QuantGranularity.CHANNELWISE)
  info = q_.OpInfo(op=q_.OperatorT(), op_name=q_.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                   op_quant_config=q_.OpQuantizationConfig(weight_tensor_config=cfg))
  p = m.gptq.get_tensor_quant_params(info, cfg, w, {"activation_tensor_qsv": {"hessian": m.rt.HbmArray(h), "num_samples": 16}})
  ref_scale = O.min_max_quant_params(w, 4, True, "CHANNELWISE")["scale"]
  assert np.array_equal(p.scale, ref_scale)
  zp = np.zeros((rows, 1), np.int8)
  ref = O.gptq_apply(w, ref_scale, zp, 4, True, None, "CHANNELWISE", hinv=O.gptq_hessian_inverse(hess, product="matmul"))
  ref_b = O.gptq_apply(w, ref_scale, zp, 4, True, None, "CHANNELWISE", hinv=O.gptq_hessian_inverse(hess_b, product="matmul"))
  floor = float((ref != ref_b).mean())
  parity_rates.note("reference noise floor: oracle FULL CHAIN [48,4096] int4, LLM-like activations, Hessian summed in another order",
                    "int_mismatch_fraction", floor, 1.0)
  parity_rates.check_default_path("LLM-like activations: get_tensor_quant_params [48,4096] int4 vs oracle FULL CHAIN (sgemm product)",
                                  np.asarray(p.quantized_data), ref, ref_b)


def test_full_chain_ill_conditioned_down_proj_d16384(m):
  """d = 16384 where the FP64-Cholesky / bf16-split TRTRI + L^-T L^-1 choice is actually stressed (every other d = 16384
  case feeds i.i.d. N(0, 1) tokens: H ~ 2 tokens I, the best-conditioned matrix there is). Activations shaped like a
  decoder's MLP hidden state: per-token scales over a decade, a per-channel mean, eight massive channels 80 x the rest,
  and a RANK-DEFICIENT TAIL -- the last 2048 channels are mixtures of the other 14336 plus 1e-3 of noise, so along 2048
  directions the Hessian is ~1e-6 of its bulk and the damp term (0.01 mean(diag), ref gptq.py:115-116) decides the
  inverse. Whole chain -- Hessian, damped inverse, OBS apply -- on 64 rows of a down_proj-shaped weight against the
  oracle's OWN chain (sgemm Hessian, FP64 Cholesky, strtri, sgemm product: ref gptq.py:100-128, 131-216), with the
  oracle's reproducibility (its Hessian summed in two halves) and BOTH sides' distance to an all-FP64 chain recorded
  beside it; H^-1 against the exact FP64 inverse of the damped matrix."""
  torch, q_ = m.torch, m.q
  d, tail, samples, per = D_BIG, 2048, 32, 512                 # 16384 tokens: no more than d (rank <= 14336 + noise)
  tokens = samples * per
  gen = torch.Generator(device="cuda").manual_seed(6006)
  base = torch.randn((tokens, d - tail), generator=gen, device="cuda")
  base = base * torch.exp(0.8 * torch.randn((tokens, 1), generator=gen, device="cuda"))       # token-correlated scale
  base = base + 0.3 * torch.randn((1, d - tail), generator=gen, device="cuda")                 # per-channel mean
  mix = torch.randn((d - tail, tail), generator=gen, device="cuda") / float(np.sqrt(d - tail))
  x = torch.cat([base, base @ mix + 1e-3 * torch.randn((tokens, tail), generator=gen, device="cuda")], dim=1)
  del base, mix
  x[:, torch.randint(0, d, (8,), generator=gen, device="cuda")] *= 80.0                      # massive channels
  x = x.contiguous()
  h = m.ops.gptq_xtx(x, 2.0 / samples)
  hinv, info = m.ops.gptq_hinv(h, 0.01)
  assert int(info.item()) == 0
  damped = _damped(torch, h)
  exact = torch.linalg.inv(damped)
  def largest_eigenvalue(a, iters=60):            # symmetric positive definite: power iteration (an SVD of order 16384 takes minutes)
    v = torch.ones((a.shape[0], 1), dtype=torch.float64, device="cuda")
    lam = 0.0
    for _ in range(iters):
      v = a @ v
      lam = float(v.norm())
      v /= lam
    return lam
  cond = largest_eigenvalue(damped) * largest_eigenvalue(exact)          # lambda_max / lambda_min (a lower estimate)
  parity_rates.note("ill-conditioned d=16384: condition number of the damped Hessian (recorded, not gated)", "condition_number",
                    cond, 1e30)
  assert cond > 1e4                                              # (N(0,1) tokens: ~10)
  err = float((hinv.double() - exact).abs().max() / exact.abs().max())
  parity_rates.note("ill-conditioned d=16384: hinv vs exact FP64 inverse", "max_rel_error", err, 7e-7)
  del damped
  # ---- the oracle's own chain on the host
  xh = x.reshape(samples, per, d).cpu().numpy()
  del x
  hess = O.gptq_hessian(xh)
  x2 = xh.reshape(-1, d)
  half = x2.shape[0] // 2
  hess_b = (2.0 / np.array(samples)) * (x2[:half].T.dot(x2[:half]) + x2[half:].T.dot(x2[half:]))
  del xh, x2
  sub = np.r_[0:64, 8000:8064, d - 64:d]
  idx = torch.from_numpy(sub).cuda()
  parity_rates.check_rel("ill-conditioned d=16384: Hessian vs oracle x.T.dot(x), 192 columns",
                         h[idx][:, idx].cpu().numpy(), hess[np.ix_(sub, sub)], 2e-6)
  ref_hinv = O.gptq_hessian_inverse(hess, product="matmul")
  ref_hinv_b = O.gptq_hessian_inverse(hess_b, product="matmul")
  del hess, hess_b
  exact32 = exact.float().cpu().numpy()
  scale_of = float(np.abs(exact32).max())
  parity_rates.note("ill-conditioned d=16384: the ORACLE's inverse (FP64 Cholesky, strtri, sgemm) vs the exact FP64 inverse (recorded)",
                    "max_rel_error", float(np.abs(ref_hinv.astype(np.float64) - exact.cpu().numpy()).max() / scale_of), 1.0)
  del exact
  rows = 64
  w = np.random.default_rng(6007).standard_normal((rows, d), dtype=np.float32) * np.float32(0.02)
  cfg = q_.TensorQuantizationConfig(num_bits=4, symmetric=True, granularity=q_.QuantGranularity.CHANNELWISE)
  info_ = q_.OpInfo(op=q_.OperatorT(), op_name=q_.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                    op_quant_config=q_.OpQuantizationConfig(weight_tensor_config=cfg))
  p = m.gptq.get_tensor_quant_params(info_, cfg, w, {"activation_tensor_qsv": {"hessian": m.rt.HbmArray(h), "num_samples": samples}})
  got = np.asarray(p.quantized_data)
  ref_scale = O.min_max_quant_params(w, 4, True, "CHANNELWISE")["scale"]
  assert np.array_equal(p.scale, ref_scale)
  zp = np.zeros((rows, 1), np.int8)
  ref = O.gptq_apply(w, ref_scale, zp, 4, True, None, "CHANNELWISE", hinv=ref_hinv)
  ref_b = O.gptq_apply(w, ref_scale, zp, 4, True, None, "CHANNELWISE", hinv=ref_hinv_b)
  truth = O.gptq_apply(w, ref_scale, zp, 4, True, None, "CHANNELWISE", hinv=exact32)        # all-FP64 Hessian and inverse
  # Recorded when this test was written (cond 1.9e5): the ORACLE's float32 steps leave its inverse 8.8e-6 from the exact one
  # (the GPU's: 5.3e-7) and 2.4e-3 of its integers differ from the all-FP64 chain's; summing its Hessian in another order
  # moves 2.0e-3 of them -- the reference does not reproduce itself below that here. The GPU's integers EQUAL the all-FP64
  # chain's (0 of 1 048 576). So the gates are: (1) the GPU against the exact-arithmetic answer under the default path's
  # fixed bound; (2) the GPU no further from the reference than the reference is from itself (two floors, capped);
  # (3) the GPU at least as close to the exact answer as the reference is.
  rates = {}
  for label, a in (("GPU", got), ("oracle", ref), ("oracle, Hessian summed in two halves", ref_b)):
    rates[label] = float((a != truth).mean())
    parity_rates.note(f"ill-conditioned d=16384 [64,16384] int4: {label} vs the all-FP64 chain (recorded)", "int_mismatch_fraction",
                      rates[label], 1.0)
  parity_rates.check_default_path("ill-conditioned d=16384: get_tensor_quant_params [64,16384] int4 vs the ALL-FP64 chain (exact Hessian, exact inverse)",
                                  got, truth)
  parity_rates.check_with_floor("ill-conditioned d=16384: get_tensor_quant_params [64,16384] int4 vs oracle FULL CHAIN (sgemm product)",
                                got, ref, ref_b, cap=5e-3, k=2.0)
  assert rates["GPU"] <= rates["oracle"]
