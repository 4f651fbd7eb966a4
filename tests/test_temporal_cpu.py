"""Host-side contract of the temporal SuperSampler mirror (no GPU): module tree -> identical state-dict
keys, shapes and seed-for-seed initialisation, against the digest the reference produced."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "temporal_model.npz")


def test_state_dict_and_seeded_initialisation_match_reference():
    from superresolution_amd.temporal import SuperSampler
    z = np.load(GOLDEN)
    torch.manual_seed(42)
    m = SuperSampler(3, 10, 4, 1)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["keys"]]
    for k, v in sd.items():
        ref = z["init." + k]
        flat = v.detach().reshape(-1)
        n = min(8, flat.numel())
        assert np.array_equal(flat[:n].numpy(), ref[:n]), k
        assert abs(flat.double().sum().item() - float(ref[-1])) <= 1e-4 * max(1.0, abs(float(ref[-1]))), k


def test_forward_without_a_gpu_fails_loudly():
    """Inference and training both run on libsrhip.so only: host tensors are refused, with or without autograd."""
    import pytest
    from superresolution_amd.temporal import SuperSampler
    m = SuperSampler(3, 10, 4, 1)
    mk = lambda *s: torch.zeros(*s)
    args = lambda: (mk(1, 3, 8, 8), mk(1, 10, 32, 32), mk(1, 10, 8, 8), mk(1, 1, 2, 8, 8), mk(1, 1, 3, 8, 8), mk(1, 1, 10, 8, 8))
    with pytest.raises(ValueError, match="GPU"):
        m(*args())
    with torch.no_grad(), pytest.raises(ValueError, match="GPU"):
        m(*args())


def _masks_golden():
    import os
    import numpy as np
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "temporal_masks.npz"))


def test_disocclusion_oracle_matches_reference_masks():
    """tests/golden/temporal_masks.npz holds the outputs of the reference's own mask functions
    (core/models/model_final.py:50-139) composed as its forward does: the oracle reproduces them exactly."""
    import numpy as np
    from oracle import oracle as orc
    z = _masks_golden()
    for t in "abc":
        m, mf, aux, ew = orc.disocclusion(z[t + ".lr_frame"], z[t + ".lr_buffer"], z[t + ".warped_frames"], z[t + ".warped_buffer"])
        assert np.array_equal(m, z[t + ".mask"])
        assert np.array_equal(mf, z[t + ".marked_frames"]) and np.array_equal(aux, z[t + ".aux"])
        assert np.array_equal(ew, z[t + ".enhance_weight"])
        assert np.abs(orc.clamp_depth(z[t + ".depth"]) - z[t + ".depth_clamped"]).max() <= 1e-6


def test_tensor_statement_of_the_masks_matches_reference():
    """The tensor-operation statement of the masks (oracle/torch_checkers.py) gives the reference's masks."""
    import numpy as np
    import torch
    from oracle import torch_checkers as T
    z = _masks_golden()
    for t in "abc":
        lr_frame, lr_buffer = torch.from_numpy(z[t + ".lr_frame"]), torch.from_numpy(z[t + ".lr_buffer"])
        wf, wb = torch.from_numpy(z[t + ".warped_frames"]), torch.from_numpy(z[t + ".warped_buffer"])
        b, f, _, h, w = wf.shape
        cur = lr_buffer.view(b, 1, -1, h, w).expand_as(wb)
        assert np.array_equal(T.compute_mask_frame(lr_frame, wf, ksize=5).numpy(), z[t + ".mask_frame"])
        assert np.array_equal(T.compute_mask_basecolor(cur[:, :, 6:9], wb[:, :, 6:9]).numpy(), z[t + ".mask_color"])
        assert np.array_equal(T.compute_mask_depth(cur[:, :, -1:], wb[:, :, -1:]).numpy(), z[t + ".mask_depth"])
        assert np.array_equal(T.compute_mask_normal(cur[:, :, 0:3], wb[:, :, 0:3]).numpy(), z[t + ".mask_normal"])
        assert np.array_equal(T.compute_mask_roughness(cur[:, :, 5:6], wb[:, :, 5:6], mark_min_value=0.02).numpy(), z[t + ".mask_rough"])
        assert np.array_equal(T.compute_mask_metallic(cur[:, :, 3:4], wb[:, :, 3:4], mark_max_value=0.95).numpy(), z[t + ".mask_metal"])
        assert np.abs(T.clamp_depth(torch.from_numpy(z[t + ".depth"])).numpy() - z[t + ".depth_clamped"]).max() <= 1e-6
