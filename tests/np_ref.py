"""Independent numpy restatement of cfar.cpp used to cross-check the C oracle (tests only).

Vectorised over the image; float32 accumulation in the reference's ascending-row order so it
matches cfar.cpp:38-47 bit for bit also for non-integer images.
"""
import numpy as np


def cfar_np(img, alg, train_hs, guard_hs, tau, k=0):
    x = np.asarray(img).astype(np.float32)
    rows, cols = x.shape
    H = train_hs + guard_hs
    mask = np.zeros((rows, cols), np.uint8)
    thr = np.zeros((rows, cols), np.float32)
    if rows - H <= H:
        return mask, thr
    rr = np.arange(H, rows - H)
    lead = np.zeros((len(rr), cols), np.float32)
    lag = np.zeros((len(rr), cols), np.float32)
    for o in range(-H, -guard_hs):          # i - row < -guard_hs, ascending i
        lead = (lead + x[rr + o]).astype(np.float32)
    for o in range(guard_hs + 1, H + 1):    # i - row > guard_hs, ascending i
        lag = (lag + x[rr + o]).astype(np.float32)
    if alg == "CA":
        s = np.zeros((len(rr), cols), np.float32)
        for o in list(range(-H, -guard_hs)) + list(range(guard_hs + 1, H + 1)):
            s = (s + x[rr + o]).astype(np.float32)
        t = tau * s.astype(np.float64) / (2.0 * train_hs)
    elif alg == "SOCA":
        t = tau * np.minimum(lead, lag).astype(np.float64) / train_hs
    elif alg == "GOCA":
        t = tau * np.maximum(lead, lag).astype(np.float64) / train_hs
    elif alg == "OS":
        offs = list(range(-H, -guard_hs)) + list(range(guard_hs + 1, H + 1))
        cells = np.stack([x[rr + o] for o in offs], 0)
        kth = np.sort(cells, axis=0)[k]
        t = tau * kth.astype(np.float64)
    else:
        raise ValueError(alg)
    mask[rr] = x[rr].astype(np.float64) > t
    thr[rr] = t.astype(np.float32)
    return mask, thr
