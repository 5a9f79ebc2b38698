"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

NumPy restatement of the forward cell recurrences AND of the hand-derived
reverse-time (BPTT) recurrences that the HIP backward kernels implement, so the
derivation itself is pinned against the reference's autograd results stored in
tests/golden/ (the reference has no explicit backward: it is the autograd
replay of snns.py:282-303 / 419-445 / 554-578 / 696-727 / 808-825 through
SpikeFunctionBoxcar.backward, snns.py:31-36).

Parity status: PINNED by tests/test_oracle_golden.py against the golden grads.

All arithmetic is float32 with one rounding per operation, written in the
reference's operation order (forward) so spikes are bit-identical to the
reference CPU path.
"""
import math

import numpy as np

f32 = np.float32

ALPHA_LIM = (f32(math.exp(-1 / 5)), f32(math.exp(-1 / 25)))
BETA_LIM = (f32(math.exp(-1 / 30)), f32(math.exp(-1 / 120)))
A_LIM = (f32(-1.0), f32(1.0))
B_LIM = (f32(0.0), f32(2.0))


def _clamp(x, lim):
    return np.minimum(np.maximum(x.astype(f32), lim[0]), lim[1])


def _inside(x, lim):
    """torch.clamp passes gradient where min <= x <= max (inclusive)."""
    return ((x >= lim[0]) & (x <= lim[1])).astype(f32)


def cell_forward(kind, Wx, p, u0, w0, s0, theta=1.0):
    """Forward of the four cells.  Returns spikes s (B,T,H) and the saved
    trajectories u (B,T,H), w (B,T,H or None)."""
    adaptive = kind in ("adLIF", "RadLIF")
    recurrent = kind in ("RLIF", "RadLIF")
    B, T, H = Wx.shape
    theta = f32(theta)
    alpha = _clamp(p["alpha"], ALPHA_LIM)
    one_m_alpha = f32(1) - alpha
    if adaptive:
        beta, a, b = _clamp(p["beta"], BETA_LIM), _clamp(p["a"], A_LIM), _clamp(p["b"], B_LIM)
    if recurrent:
        Vm = p["V"].astype(f32).copy()
        np.fill_diagonal(Vm, 0)
    u, s = u0.astype(f32), s0.astype(f32)
    w = w0.astype(f32) if adaptive else None
    S = np.empty((B, T, H), f32)
    U = np.empty((B, T, H), f32)
    W = np.empty((B, T, H), f32) if adaptive else None
    for t in range(T):
        drive = Wx[:, t, :]
        if adaptive:
            w = (beta * w + a * u) + b * s
        if recurrent:
            drive = drive + s @ Vm
        if adaptive:
            drive = drive - w
            W[:, t] = w
        u = alpha * (u - s) + one_m_alpha * drive
        s = ((u - theta) > 0).astype(f32)
        U[:, t], S[:, t] = u, s
    return S, U, W


def cell_backward(kind, g_s, Wx, p, u0, w0, s0, U, W, theta=1.0):
    """Reverse-time recurrences.  With du_{T+1}=dw_{T+1}=0, for t=T..1:

        ds_t  = g_t - alpha*du_{t+1} + [b*dw_{t+1}] + [((1-alpha)*du_{t+1}) @ V^T]
        du_t  = ds_t*box(u_t-theta) + alpha*du_{t+1} + [a*dw_{t+1}]
        dw_t  = beta*dw_{t+1} - (1-alpha)*du_t
        dWx_t = (1-alpha)*du_t
        dalpha += sum_b du_t*((u_{t-1}-s_{t-1}) - drive_t)
        dbeta  += sum_b dw_t*w_{t-1};  da += sum_b dw_t*u_{t-1};  db += sum_b dw_t*s_{t-1}
        dV     += s_{t-1}^T @ dWx_t;   zero diag(dV) at the end

    Returns dict(dWx, dalpha[, dbeta, da, db][, dV]) with clamp gating applied."""
    adaptive = kind in ("adLIF", "RadLIF")
    recurrent = kind in ("RLIF", "RadLIF")
    B, T, H = g_s.shape
    theta = f32(theta)
    alpha = _clamp(p["alpha"], ALPHA_LIM)
    oma = f32(1) - alpha
    if adaptive:
        beta, a, b = _clamp(p["beta"], BETA_LIM), _clamp(p["a"], A_LIM), _clamp(p["b"], B_LIM)
    if recurrent:
        Vm = p["V"].astype(f32).copy()
        np.fill_diagonal(Vm, 0)

    du_next = np.zeros((B, H), f32)
    dw_next = np.zeros((B, H), f32)
    dWx = np.empty((B, T, H), f32)
    dalpha = np.zeros(H, np.float64)
    dbeta = np.zeros(H, np.float64)
    da = np.zeros(H, np.float64)
    db = np.zeros(H, np.float64)
    dV = np.zeros((H, H), np.float64)
    for t in range(T - 1, -1, -1):
        u_t = U[:, t]
        u_prev = U[:, t - 1] if t > 0 else u0.astype(f32)
        s_prev = ((U[:, t - 1] - theta) > 0).astype(f32) if t > 0 else s0.astype(f32)
        ds = g_s[:, t] - alpha * du_next
        if adaptive:
            ds = ds + b * dw_next
        if recurrent:
            ds = ds + (oma * du_next) @ Vm.T
        x = u_t - theta
        du = np.where((x <= -0.5) | (x > 0.5), f32(0), ds) + alpha * du_next  # masked assignment, snns.py:33-35
        if adaptive:
            du = du + a * dw_next
            dw = beta * dw_next - oma * du
        dWx_t = oma * du
        dWx[:, t] = dWx_t
        # drive_t recovered from the forward relation u_t = alpha*q + (1-alpha)*drive_t
        q = u_prev - s_prev
        drive = Wx[:, t].astype(f32)
        if recurrent:
            drive = drive + s_prev @ Vm
        if adaptive:
            drive = drive - W[:, t]
        dalpha += (du * (q - drive)).sum(0, dtype=np.float64)
        if adaptive:
            w_prev = W[:, t - 1] if t > 0 else w0.astype(f32)
            dbeta += (dw * w_prev).sum(0, dtype=np.float64)
            da += (dw * u_prev).sum(0, dtype=np.float64)
            db += (dw * s_prev).sum(0, dtype=np.float64)
            dw_next = dw
        if recurrent:
            dV += s_prev.astype(np.float64).T @ dWx_t.astype(np.float64)
        du_next = du
    out = {"dWx": dWx, "dalpha": (dalpha * _inside(p["alpha"], ALPHA_LIM)).astype(f32)}
    if adaptive:
        out["dbeta"] = (dbeta * _inside(p["beta"], BETA_LIM)).astype(f32)
        out["da"] = (da * _inside(p["a"], A_LIM)).astype(f32)
        out["db"] = (db * _inside(p["b"], B_LIM)).astype(f32)
    if recurrent:
        np.fill_diagonal(dV, 0)
        out["dV"] = dV.astype(f32)
    return out


def readout_forward(Wx, alpha_raw, u0):
    """snns.py:808-825.  Returns out (B,C) and saved u (B,T,C)."""
    B, T, C = Wx.shape
    alpha = _clamp(alpha_raw, ALPHA_LIM)
    oma = f32(1) - alpha
    u = u0.astype(f32)
    out = np.zeros((B, C), f32)
    U = np.empty((B, T, C), f32)
    for t in range(T):
        u = alpha * u + oma * Wx[:, t]
        U[:, t] = u
        e = np.exp(u - u.max(1, keepdims=True))
        out = out + (e / e.sum(1, keepdims=True)).astype(f32)
    return out, U


def readout_backward(g_out, Wx, alpha_raw, u0, U):
    """du_t = alpha*du_{t+1} + p_t*(g - <p_t,g>),  p_t = softmax(u_t);
    dWx_t = (1-alpha)*du_t;  dalpha += sum_b du_t*(u_{t-1} - Wx_t)."""
    B, T, C = Wx.shape
    alpha = _clamp(alpha_raw, ALPHA_LIM)
    oma = f32(1) - alpha
    du = np.zeros((B, C), f32)
    dWx = np.empty((B, T, C), f32)
    dalpha = np.zeros(C, np.float64)
    for t in range(T - 1, -1, -1):
        u = U[:, t]
        e = np.exp(u - u.max(1, keepdims=True))
        pr = (e / e.sum(1, keepdims=True)).astype(f32)
        dsm = pr * (g_out - (pr * g_out).sum(1, keepdims=True))
        du = alpha * du + dsm
        dWx[:, t] = oma * du
        u_prev = U[:, t - 1] if t > 0 else u0.astype(f32)
        dalpha += (du * (u_prev - Wx[:, t])).sum(0, dtype=np.float64)
    return {"dWx": dWx, "dalpha": (dalpha * _inside(alpha_raw, ALPHA_LIM)).astype(f32)}
