# coding: utf-8
"""ORACLE #2 (test infrastructure): an independent numpy-fp64 restatement.

Written separately from oracle/ref_torch.py (different decomposition: explicit
per-head einsums, closed-form cross entropy, per-sentence beam bookkeeping,
full-prefix recomputation instead of caches) so that agreement between the
two pins the restatement -- TF1 itself cannot run here (PARITY UNPINNED by the
reference's own tests; see oracle/ref_torch.py header).

Forward only: loss / per-sentence loss / logits, and beam search that
recomputes the whole prefix each step (the reference's ``search_mode="dev"``
semantics, search.py:129-142, transformer.py:277-281).
"""

import math

import numpy as np

F32_MIN = np.finfo(np.float32).min
EPS = 1e-8   # utils/dtype.py:14
INF = 1e8    # utils/dtype.py:15


def _ln(x, g, b):
    # func.py:300-303
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return g * (x - mu) / np.sqrt(var + EPS) + b


def _softmax(z):
    z = z - z.max(-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(-1, keepdims=True)


def _timing(L, C, pos0=0):
    # func.py:341-369
    nts = C // 2
    inc = math.log(1.0e4) / (nts - 1.0)
    inv = np.exp(-inc * np.arange(nts))
    p = (np.arange(L) + pos0)[:, None] * inv[None, :]
    out = np.zeros((L, C))
    out[:, :nts] = np.sin(p)
    out[:, nts:2 * nts] = np.cos(p)
    return out


def _keep_cols(ids):
    # utils/util.py:274-287
    keep = (ids != 0).any(0)
    keep[0] = True
    return ids[:, keep]


def _mha(xq, xkv, P, scope, nh, key_bias, causal, rpr, max_rel, fuse_mask=None):
    """func.py:164-286.  xq [B,Lq,H]; xkv None (self) or [B,Lk,H];
    key_bias [B,Lk] additive (0 / -INF) or None."""
    H = xq.shape[-1]
    d = H // nh
    s = scope + "/dot_attention/"
    if xkv is None:
        qkv = xq @ P[s + "qkv_map/W_0_0"] + P[s + "qkv_map/b_0"]
        q, k, v = qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:]
    else:
        q = xq @ P[s + "q_map/W_0_0"] + P[s + "q_map/b_0"]
        k = xkv @ P[s + "k_map/W_0_0"] + P[s + "k_map/b_0"]
        v = xkv @ P[s + "v_map/W_0_0"] + P[s + "v_map/b_0"]
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    out = np.zeros((B, Lq, H))
    if rpr:
        ii = np.arange(Lq)[:, None]
        jj = np.arange(Lk)[None, :]
        ridx = np.clip(ii - jj, -max_rel, max_rel) + max_rel   # modules/rpr.py:62-75
        Rk = P[s + "rpr_keys/embeddings"][ridx]      # [Lq,Lk,d]
        Rv = P[s + "rpr_values/embeddings"][ridx]
    for h in range(nh):
        qh = q[..., h * d:(h + 1) * d] * d ** -0.5   # func.py:222
        kh = k[..., h * d:(h + 1) * d]
        vh = v[..., h * d:(h + 1) * d]
        lg = np.einsum("bqc,bkc->bqk", qh, kh)
        if rpr:
            lg = lg + np.einsum("bqc,qkc->bqk", qh, Rk)
        if key_bias is not None:
            lg = lg + key_bias[:, None, :]
        if causal:
            lg = lg + (-INF) * (1.0 - np.tril(np.ones((Lq, Lk))))[None]
        w = _softmax(lg)
        oh = np.einsum("bqk,bkc->bqc", w, vh)
        if rpr:
            oh = oh + np.einsum("bqk,qkc->bqc", w, Rv)
        out[..., h * d:(h + 1) * d] = oh
    if fuse_mask is not None:
        # func.py:258-275: averaged v_map(query) over valid positions j <= i (zero on pad rows)
        m = fuse_mask
        vq = xq @ P[s + "v_map/W_0_0"] + P[s + "v_map/b_0"]
        cm = np.cumsum(m, axis=1)
        out = out + np.cumsum(vq * m[..., None], axis=1) / np.maximum(cm, 1.0)[..., None] * m[..., None]
    return out @ P[s + "o_map/W_0_0"] + P[s + "o_map/b_0"]


def _ffn(x, P, scope):
    # func.py:327-338
    s = scope + "/ffn_layer/"
    h = np.maximum(x @ P[s + "enlarge/W_0_0"] + P[s + "enlarge/b_0"], 0.0)
    return h @ P[s + "output/W_0_0"] + P[s + "output/b_0"]


def _lnp(x, P, scope):
    return _ln(x, P[scope + "/layer_norm/scale"], P[scope + "/layer_norm/offset"])


def _emb(hp, which):
    if hp.shared_source_target_embedding:
        return "embedding"
    if which == "soft":
        return "tgt_embedding" if hp.shared_target_softmax_embedding else "softmax_embedding"
    return which + "_embedding"


def encode(src, hp, P, model):
    """transformer.py:15-84."""
    H = hp.hidden_size
    src = _keep_cols(src)
    m = (src != 0).astype(np.float64)
    x = P[_emb(hp, "src")][src] * math.sqrt(H) + P["bias"] + _timing(src.shape[1], H)[None]
    kb = (1.0 - m) * -INF
    rpr = model == "transformer_rpr"
    for l in range(hp.num_encoder_layer):
        s = "encoder/layer_%d" % l
        x = _lnp(x + _mha(x, None, P, s + "/self_attention", hp.num_heads, kb, False, rpr,
                          hp.max_relative_position), P, s + "/self_attention")
        x = _lnp(x + _ffn(x, P, s + "/feed_forward"), P, s + "/feed_forward")
    return x, m


def decode_full(tgt, enc, smask, hp, P, model):
    """Training-path decoder (transformer.py:87-181 / transformer_aan.py:120-223):
    returns the final features [B,Lt,H] for shifted inputs."""
    H = hp.hidden_size
    B, Lt = tgt.shape
    m = (tgt != 0).astype(np.float64)
    e = P[_emb(hp, "tgt")][tgt] * math.sqrt(H) + P["bias"]
    x = np.zeros((B, Lt, H))
    x[:, 1:] = e[:, :-1]                       # transformer.py:108-110
    x = x + _timing(Lt, H)[None]
    kb = (1.0 - smask) * -INF
    rpr = model == "transformer_rpr"
    for l in range(hp.num_decoder_layer):
        s = "decoder/layer_%d" % l
        if model == "transformer_fuse":
            # transformer_fuse.py:131-160
            x = _lnp(x + _mha(x, enc, P, s + "/fuse_attention", hp.num_heads, kb, False, False,
                              hp.max_relative_position, fuse_mask=m), P, s + "/fuse_attention")
            x = _lnp(x + _ffn(x, P, s + "/feed_forward"), P, s + "/feed_forward")
            continue
        if model == "transformer_aan":
            a = s + "/average_attention"
            if hp.aan_mask:
                # func.py:390-398: average over valid j<=i, zero on pad rows
                cm = np.cumsum(m, axis=1)
                y = np.cumsum(x * m[..., None], axis=1) / np.maximum(cm, 1.0)[..., None]
                y = y * m[..., None]
            else:
                # transformer_aan.py:103-108
                cm = np.cumsum(m, axis=1)
                cm = np.where(cm <= 0, 1.0, cm)
                y = np.cumsum(x, axis=1) / cm[..., None]
            if hp.use_ffn:
                y = _ffn(y, P, a)
            z = np.concatenate([x, y], -1) @ P[a + "/z_project/W_0_0"] + P[a + "/z_project/b_0"]
            gi, gf = z[..., :H], z[..., H:]
            y = x / (1 + np.exp(-gi)) + y / (1 + np.exp(-gf))
            x = _lnp(x + y, P, a)
        else:
            x = _lnp(x + _mha(x, None, P, s + "/self_attention", hp.num_heads, None, True, rpr,
                              hp.max_relative_position), P, s + "/self_attention")
        x = _lnp(x + _mha(x, enc, P, s + "/cross_attention", hp.num_heads, kb, False, rpr,
                          hp.max_relative_position), P, s + "/cross_attention")
        x = _lnp(x + _ffn(x, P, s + "/feed_forward"), P, s + "/feed_forward")
    return x, m


def loss_fn(src, tgt, hp, P, model, label_smooth=None):
    """transformer.py:182-216 with the closed-form smoothed cross entropy
    ce = lse - p*z_gold - q*(sum z - z_gold) - normaliser (util.py:88-103)."""
    P = {k: np.asarray(v, dtype=np.float64) for k, v in P.items()}
    ls = hp.label_smooth if label_smooth is None else label_smooth
    enc, smask = encode(src, hp, P, model)
    tgt = _keep_cols(tgt)
    feat, m = decode_full(tgt, enc, smask, hp, P, model)
    E = P[_emb(hp, "soft")]
    z = feat @ E.T                       # [B,Lt,V]
    V = z.shape[-1]
    zmax = z.max(-1, keepdims=True)
    lse = (zmax + np.log(np.exp(z - zmax).sum(-1, keepdims=True)))[..., 0]
    zg = np.take_along_axis(z, tgt[..., None], -1)[..., 0]
    if 0.0 < ls < 1.0:
        n = V - 1.0
        p, q = 1.0 - ls, ls / n
        ce = lse - p * zg - q * (z.sum(-1) - zg)
        norm = -(np.float32(p) * np.log(np.float32(p)) +
                 np.float32(n) * np.float32(q) * np.log(np.float32(q) + np.float32(1e-20)))
        ce = ce - float(norm)
    else:
        ce = lse - zg
    per = (ce * m).sum(-1) / m.sum(-1)
    return {"loss": per.mean(), "per_sample_loss": per, "logits": z.reshape(-1, V)}


def _next_logits(src, prefix, hp, P, model):
    """logits for the next symbol given generated prefix [N,t] (no BOS):
    dev-mode target = prefix + dummy token 1 (search.py:139-140)."""
    enc, smask = encode(src, hp, P, model)
    tgt = np.concatenate([prefix, np.ones((prefix.shape[0], 1), dtype=prefix.dtype)], 1)
    # note: the reference's remove_invalid_seq would drop generated all-pad
    # columns here; prefixes in tests never contain id 0 in every row.
    feat, _ = decode_full(tgt, enc, smask, hp, P, model)
    return feat[:, -1, :] @ P[_emb(hp, "soft")].T


def beam_search(src, hp, P, model):
    """search.py:19-275 with per-sentence bookkeeping (python lists)."""
    P = {k: np.asarray(v, dtype=np.float64) for k, v in P.items()}
    K, alpha = hp.beam_size, hp.decode_alpha
    eos, pad = hp.tgt_vocab.eos(), hp.tgt_vocab.pad()
    B = src.shape[0]
    f32 = np.float32
    src_len = (src != 0).sum(1).astype(f32)
    max_len = src_len + f32(hp.decode_length)
    alive_seq = [[[] for _ in range(K)] for _ in range(B)]
    alive_lp = np.tile(np.array([0.] + [F32_MIN] * (K - 1), dtype=f32), (B, 1))
    alive_sc = np.zeros((B, K), dtype=f32)
    fin_seq = [[[] for _ in range(K)] for _ in range(B)]
    fin_sc = np.full((B, K), F32_MIN, dtype=f32)
    fin_fl = np.zeros((B, K), dtype=bool)
    t = 0
    while True:
        maxpen = np.power((f32(5.) + max_len) / f32(6.), f32(alpha)).astype(f32)
        best_alive = alive_lp[:, 0] / maxpen
        worst = (fin_sc * fin_fl.astype(f32)).min(1) + (f32(1.) - fin_fl.any(1).astype(f32)) * F32_MIN
        if (worst > best_alive).all() or not (t < max_len.astype(np.int32)).any():
            break
        prefix = np.array([[alive_seq[b][k] for k in range(K)] for b in range(B)],
                          dtype=np.int64).reshape(B * K, t)
        src_t = np.repeat(src, K, axis=0)
        lg = _next_logits(src_t, prefix, hp, P, model).astype(f32) / f32(hp.beam_search_temperature)
        mx = lg.max(-1, keepdims=True)
        lp = (lg - (mx + np.log(np.exp(lg - mx).sum(-1, keepdims=True)))).astype(f32)
        V = lp.shape[-1]
        if t < 1:
            lp[:, eos] = lp[:, eos] + f32(-INF)
        lp = lp.reshape(B, K, V)
        pen = np.power(f32((f32(5.) + f32(t + 1)) / f32(6.)), f32(alpha)).astype(f32)
        n_alive_seq, n_fin_seq = [], []
        n_alive_lp = np.zeros_like(alive_lp)
        n_alive_sc = np.zeros_like(alive_sc)
        n_fin_sc = np.zeros_like(fin_sc)
        n_fin_fl = np.zeros_like(fin_fl)
        for b in range(B):
            cur = ((alive_lp[b][:, None] + lp[b]).astype(f32) / pen).astype(f32).reshape(-1)
            order = np.argsort(-cur, kind="stable")[:2 * K]
            cand = []
            for idx in order:
                kb, sym = int(idx) // V, int(idx) % V
                done = (sym == eos) or (t >= int(max_len[b]))
                cand.append((cur[idx], alive_seq[b][kb] + [sym], done))
            a_sc = np.array([c[0] + (F32_MIN if c[2] else f32(0.)) for c in cand], dtype=f32)
            a_ord = np.argsort(-a_sc, kind="stable")[:K]
            n_alive_seq.append([cand[i][1] for i in a_ord])
            n_alive_sc[b] = a_sc[a_ord]
            n_alive_lp[b] = (a_sc[a_ord] * pen).astype(f32)
            f_sc = np.array(list(fin_sc[b]) +
                            [c[0] + (f32(0.) if c[2] else F32_MIN) for c in cand], dtype=f32)
            f_fl = list(fin_fl[b]) + [c[2] for c in cand]
            f_sq = [s + [pad] for s in fin_seq[b]] + [c[1] for c in cand]
            f_ord = np.argsort(-f_sc, kind="stable")[:K]
            n_fin_seq.append([f_sq[i] for i in f_ord])
            n_fin_sc[b] = f_sc[f_ord]
            n_fin_fl[b] = [f_fl[i] for i in f_ord]
        alive_seq, alive_lp, alive_sc = n_alive_seq, n_alive_lp, n_alive_sc
        fin_seq, fin_sc, fin_fl = n_fin_seq, n_fin_sc, n_fin_fl
        t += 1
    seqs = np.zeros((B, K, t), dtype=np.int64)
    scores = np.zeros((B, K), dtype=f32)
    for b in range(B):
        use_fin = fin_fl[b].any()
        for k in range(K):
            s = fin_seq[b][k] if use_fin else alive_seq[b][k]
            s = (s + [0] * t)[:t] if use_fin else s
            seqs[b, k, :len(s)] = s
            scores[b, k] = fin_sc[b, k] if use_fin else alive_sc[b, k]
    return {"seq": seqs, "score": scores, "steps": t}
