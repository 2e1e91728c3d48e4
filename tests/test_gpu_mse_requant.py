"""MSE scale + quantize in one launch (mi355q_mse_requant_f32): the scales and integers of the two-kernel route
(mi355q_mse_scale_f32 + mi355q_quantize_f32, MI355Q_MSE_TWO_KERNELS=1) and of the oracle (ref mse.py:100-128), bit for bit,
over unit lengths whose pairwise tree is complete (one kernel) and lengths that fall back to the two kernels."""
import os

import numpy as np
import pytest

from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as ge
  ge.build()
  from mi355q import ops
  return dict(torch=torch, ops=ops)


def _run(g, w, bits, two_kernels):
  torch, ops = g["torch"], g["ops"]
  if two_kernels:
    os.environ["MI355Q_MSE_TWO_KERNELS"] = "1"
  try:
    xd = torch.from_numpy(w.reshape(-1)).cuda()
    scale, q = ops.mse_requant(xd, w.shape[0], w.shape[1], {8: 0.05408, 4: 0.37755}[bits], bits, bits >= 8)
    torch.cuda.synchronize()
    return scale.cpu().numpy(), q.cpu().numpy().reshape(w.shape)
  finally:
    os.environ.pop("MI355Q_MSE_TWO_KERNELS", None)


@pytest.mark.parametrize("unit_len", [4, 100, 128, 256, 1000, 1024, 4096, 8192, 8192 + 4096, 11008])
@pytest.mark.parametrize("bits", [4, 8])
def test_one_launch_equals_two_kernels_and_the_oracle(g, unit_len, bits):
  rng = np.random.default_rng(unit_len + bits)
  units = 37 if unit_len > 2000 else 301
  w = (rng.standard_normal((units, unit_len)) * 0.03).astype(np.float32)
  w[1] = 0.0                                   # scale 0: x / 0 -> NaN / inf, the reference's integers for it
  w[2, ::3] = -0.0
  w[3, 0] = 250.0                              # one outlier: everything else rounds to 0, the outlier clips
  with np.errstate(all="ignore"):
    ref = O.mse_quant_params(w, bits, "CHANNELWISE")
  s1, q1 = _run(g, w, bits, False)
  s2, q2 = _run(g, w, bits, True)
  assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32)) and np.array_equal(q1, q2)
  assert np.array_equal(s1.view(np.uint32), np.asarray(ref["scale"], np.float32).reshape(-1).view(np.uint32))
  assert np.array_equal(q1, ref["quantized_data"])


def test_layer_sized_weight(g):
  torch = g["torch"]
  w = (torch.randn((4096, 4096), generator=torch.Generator(device="cuda").manual_seed(3), device="cuda") * 0.02).cpu().numpy()
  s1, q1 = _run(g, w, 4, False)
  s2, q2 = _run(g, w, 4, True)
  assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32)) and np.array_equal(q1, q2)
