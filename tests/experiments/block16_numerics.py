import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import workloads as wl
f32 = np.float32

def tdf2_f64(x, coef, s0=None):
    y = np.asarray(x, np.float64).copy(); ns = coef.shape[0]
    st = np.zeros((ns, 2)) if s0 is None else s0.copy()
    for j in range(ns):
        b0,b1,b2,a1,a2 = [float(v) for v in coef[j]]
        d0, d1 = st[j]
        for i in range(len(y)):
            xx = y[i]; yy = b0*xx + d0
            d0 = b1*xx + d1 + a1*yy; d1 = b2*xx + a2*yy; y[i] = yy
        st[j] = (d0, d1)
    return y, st

def fma32(a, b, c):  # emulate fmaf
    return f32(np.float64(a)*np.float64(b) + np.float64(c))

def tdf2_f32(x, coef, s0=None):
    y = np.asarray(x, f32).copy(); ns = coef.shape[0]
    st = np.zeros((ns, 2), f32) if s0 is None else s0.astype(f32).copy()
    for j in range(ns):
        b0,b1,b2,a1,a2 = [f32(v) for v in coef[j]]
        d0, d1 = st[j]
        for i in range(len(y)):
            xx = y[i]
            tq = fma32(b1, xx, d1); u = f32(b2*xx); yy = fma32(b0, xx, d0)
            d0 = fma32(a1, yy, tq); d1 = fma32(a2, yy, u); y[i] = yy
        st[j] = (d0, d1)
    return y, st

def system16(coef):
    """A (2ns x 2ns), B (2ns): s' = A s + B x for the TDF-II cascade, float64."""
    ns = coef.shape[0]; n = 2*ns
    def step(s, x):
        s = s.copy(); u = x
        for j in range(ns):
            b0,b1,b2,a1,a2 = [float(v) for v in coef[j]]
            d0, d1 = s[2*j], s[2*j+1]
            y = b0*u + d0
            s[2*j] = b1*u + d1 + a1*y; s[2*j+1] = b2*u + a2*y; u = y
        return s
    A = np.zeros((n, n)); 
    for i in range(n):
        e = np.zeros(n); e[i] = 1; A[:, i] = step(e, 0.0)
    B = step(np.zeros(n), 1.0)
    return A, B

def run(ch, coef, x, L=16):
    ns = coef.shape[0]; n = 2*ns
    A, B = system16(coef)
    G = np.zeros((n, L)); v = B.copy()
    for k in range(L-1, -1, -1):
        G[:, k] = v; v = A @ v
    P = np.linalg.matrix_power(A, L)
    G32, P32 = G.astype(f32), P.astype(f32)
    N = len(x); nchunk = N // L
    # exact states at chunk boundaries
    yex, _ = tdf2_f64(x, coef)
    sex = np.zeros((nchunk+1, n)); s = np.zeros(n)
    for c in range(nchunk):
        for i in range(L):
            s = A @ s + B*float(x[c*L+i])
        sex[c+1] = s
    # fp32 block recurrence: s' = fmachain(G x) then (P s) accumulate (MFMA: acc over k sequentially)
    s32 = np.zeros((nchunk+1, n), f32)
    for c in range(nchunk):
        acc = np.zeros(n, np.float64)
        xs = x[c*L:(c+1)*L].astype(f32)
        a = np.zeros(n, f32)
        for k in range(L):
            a = (G32[:, k].astype(np.float64)*np.float64(xs[k]) + a.astype(np.float64)).astype(f32)
        for i in range(n):
            a = (P32[:, i].astype(np.float64)*np.float64(s32[c, i]) + a.astype(np.float64)).astype(f32)
        s32[c+1] = a
    serr = np.abs(s32 - sex).max(axis=0) / (np.abs(sex).max(axis=0) + 1e-300)
    # outputs: exact fp32 recurrence per chunk from the block start states
    y = np.zeros(N, f32)
    for c in range(nchunk):
        yy, _ = tdf2_f32(x[c*L:(c+1)*L], coef, s32[c].reshape(ns, 2))
        y[c*L:(c+1)*L] = yy
    y32, _ = tdf2_f32(x, coef)
    peak = np.abs(yex).max()
    return dict(ch=ch, cond=np.abs(P).max(), gpu_vs_exact=np.abs(y-yex).max()/peak, noise=np.abs(y32-yex).max()/peak,
                gpu_vs_ref32=np.abs(y.astype(np.float64)-y32).max()/peak, state_err=serr.max())

coef, fc = wl.c2_coefficients(1024)
x = wl.c2_input(1024, 4096)[0]
order = np.argsort(fc)
for ch in list(order[:4]) + list(order[500:502]) + list(order[-2:]):
    r = run(ch, coef[ch], x[ch][:2048])
    print("fc %8.1f  maxP %9.3g  gpu_vs_exact %.2e  noise %.2e  gpu_vs_ref32 %.2e  state_err %.2e" % (fc[ch], r['cond'], r['gpu_vs_exact'], r['noise'], r['gpu_vs_ref32'], r['state_err']))
