"""Whole-volume self-SR inference (mirror of the reference's utils/sr_utils.py, hot-path part only).

apply_to_vol_flavr (ref utils/sr_utils.py:102-135) slides a 4-slice window over the through-plane axis and
calls the network once per window at batch 1.  The windows are independent, so here they are gathered on
the device and pushed through the network in large batches (same kernels as training, far fewer launches,
full CUs at 16-multiple slice sizes); the result tensor is identical in layout and values."""
import torch


def _window_indices(S):
    """Slice indices of every window exactly as the reference builds them (-1 = zero slice)."""
    wins = []
    for st in range(0, S - 1):
        if st == 0:
            src = list(range(0, min(3, S)))
            idx = [-1] * (4 - len(src)) + src
        elif st == S - 2:
            src = list(range(st - 1, S))
            idx = src + [-1] * (4 - len(src))
        else:
            idx = list(range(st - 1, st + 3))
        wins.append(idx)
    return wins


def apply_to_vol_flavr(model, image, pred_out_idx=None, window_batch=32):
    """image (slices, C, X, Y) -> (4*(slices-1), C_out, Y, X) on the CPU, as the reference returns it."""
    ori_x, ori_y = image.shape[2], image.shape[3]
    pad_x, pad_y = (-ori_x) % 16, (-ori_y) % 16
    if pad_x or pad_y:
        image = torch.nn.functional.pad(image, (0, pad_y, 0, pad_x))
    S = image.shape[0]
    wins = _window_indices(S)
    if not wins:
        raise ValueError("apply_to_vol_flavr needs at least two slices")
    src = torch.cat([image, torch.zeros_like(image[:1])], 0)            # index S (= -1) is the zero slice
    idx = torch.tensor(wins, device=image.device) % (S + 1)             # (n_windows, 4)
    outs = []
    for i in range(0, len(wins), window_batch):
        b = src[idx[i:i + window_batch]]                                # (b, 4, C, X, Y)
        batch_input = b.permute(0, 2, 1, 4, 3).contiguous()             # (b, C, 4, Y, X): a fresh tensor (the model
        with torch.inference_mode():                                    # rewrites channel 0 of its input in place)
            sr = model(batch_input)
            if pred_out_idx is not None and isinstance(sr, tuple):
                sr = sr[pred_out_idx]
        outs.append(sr.detach()[:, :, :, :ori_y, :ori_x])
    res = torch.cat(outs, 0)                                            # (n_windows, C_out, 4, Y, X)
    res = res.permute(0, 2, 1, 3, 4).reshape(-1, res.shape[1], ori_y, ori_x)
    return res.cpu()
