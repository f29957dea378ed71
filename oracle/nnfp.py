"""Oracle (TEST INFRASTRUCTURE): fingerprint encoder.  PARITY UNPINNED (see oracle/__init__.py).

Follows model/fp/nnfp.py:20-231 of the reference.  The arithmetic lives in
tensorflow/keras layers (called at nnfp.py:48-79, 135-137, 151, 155, 218, 229);
their published semantics restated here:

  * keras `Conv2D(padding='SAME')`: cross-correlation, kernel (kh,kw,Cin,Cout),
    bias; TF SAME padding: out = ceil(in/s), pad_total = max((out-1)*s+k-in, 0),
    pad_before = pad_total//2 (the odd element goes AFTER).
  * keras `ELU()`: x if x>0 else exp(x)-1.
  * keras `LayerNormalization(axis=(1,2,3))`: mean / biased variance over all of
    (F,T,C) per sample, epsilon=1e-3, gamma/beta of shape (F,T,C).
  * keras `Dense`: x@W+b.  `tf.math.l2_normalize`: x*rsqrt(max(sum(x^2),1e-12)).
  * the alternates of MODEL.BN (nnfp.py:63-71): keras `LayerNormalization(axis=-1)` ('layer_norm1d'): mean / biased variance
    over the channels of one position, epsilon=1e-3, gamma/beta of shape (C,); keras `BatchNormalization(axis=-1)` (any other
    string): the reference calls the model as `m_fp(feat)` with no `training` argument everywhere (trainer.py:44, 60, 73, 86;
    generate.py:88) and never sets the learning phase, so keras resolves `training` to False (Layer.__call__: an unset
    `training` falls back to the learning phase, 0 outside `fit`) and the layer applies
    (x - moving_mean) / sqrt(moving_variance + 1e-3) * gamma + beta with the moving statistics it was initialised with
    (0 / 1) or restored from a checkpoint; they are never updated.
"""
import numpy as np

FRONT_HIDDEN_CH = [128, 128, 256, 256, 512, 512, 1024, 1024]        # nnfp.py:193
FRONT_STRIDES = [[(1, 2), (2, 1)], [(1, 2), (2, 1)], [(1, 2), (2, 1)], [(1, 2), (2, 1)],
                 [(1, 1), (2, 1)], [(1, 2), (2, 1)], [(1, 1), (2, 1)], [(1, 2), (2, 1)]]  # nnfp.py:194-197
LN_EPS = 1e-3


def same_pad(n_in, k, s):
    """TF 'SAME' geometry along one axis -> (n_out, pad_before, pad_after)."""
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    return n_out, total // 2, total - total // 2


def conv_geometry(input_shape=(256, 32, 1), hidden_ch=None, strides=None):
    """Per-conv geometry list following ConvLayer (nnfp.py:43-61) x8 (nnfp.py:210-216).

    Each entry: dict(name, axis ('T' for 1x3, 'F' for 3x1), in=(F,T,C), out=(F,T,C),
    stride, pad=(before, after)).
    """
    hidden_ch = hidden_ch or FRONT_HIDDEN_CH
    strides = strides or FRONT_STRIDES
    F, T, C = input_shape
    geo = []
    for i, (ch, st) in enumerate(zip(hidden_ch, strides)):
        # conv 1x3: kernel spans T; strides[0] = (sF, sT)
        sF, sT = st[0]
        Fo, pfb, pfa = same_pad(F, 1, sF)
        To, ptb, pta = same_pad(T, 3, sT)
        geo.append(dict(name=f'b{i}.conv1x3', axis='T', inp=(F, T, C), out=(Fo, To, ch),
                        stride=(sF, sT), pad=(ptb, pta)))
        F, T, C = Fo, To, ch
        sF, sT = st[1]
        Fo, pfb, pfa = same_pad(F, 3, sF)
        To, ptb, pta = same_pad(T, 1, sT)
        geo.append(dict(name=f'b{i}.conv3x1', axis='F', inp=(F, T, C), out=(Fo, To, ch),
                        stride=(sF, sT), pad=(pfb, pfa)))
        F, T, C = Fo, To, ch
    return geo


def count_params(input_shape=(256, 32, 1), emb_sz=128, fc_unit=(32, 1)):
    """Known answer: (256,63,1) -> 19,224,576 (nnfp.py:271)."""
    geo = conv_geometry(input_shape)
    conv = sum(3 * g['inp'][2] * g['out'][2] + g['out'][2] for g in geo)
    ln = sum(2 * g['out'][0] * g['out'][1] * g['out'][2] for g in geo)
    flat = geo[-1]['out'][0] * geo[-1]['out'][1] * geo[-1]['out'][2]
    sl = flat // emb_sz
    div = emb_sz * (sl * fc_unit[0] + fc_unit[0] + fc_unit[0] * fc_unit[1] + fc_unit[1])
    return dict(conv=conv, ln=ln, divenc=div, total=conv + ln + div)


def init_weights(seed=0, input_shape=(256, 32, 1), emb_sz=128, randomize_affine=False,
                 dtype=np.float32):
    """Keras-default initialisation (glorot-uniform kernels, zero biases, LN gamma=1
    beta=0; nnfp.py:48-59, 135-137).  `randomize_affine=True` perturbs biases and
    LN gamma/beta so that parity tests exercise those terms.

    Returns dict name -> ndarray with the reference's variable shapes:
      conv{j}.kernel (kh,kw,Cin,Cout), conv{j}.bias (Cout,), ln{j}.gamma/.beta (F,T,C)
      for j in 0..15 (even j = 1x3, odd j = 3x1), and
      div.w1 (Q,S,32), div.b1 (Q,32), div.w2 (Q,32,1), div.b2 (Q,1).
    """
    rng = np.random.default_rng(seed)
    geo = conv_geometry(input_shape)
    w = {}
    for j, g in enumerate(geo):
        cin, cout = g['inp'][2], g['out'][2]
        kh, kw = (1, 3) if g['axis'] == 'T' else (3, 1)
        lim = np.sqrt(6.0 / (3 * cin + 3 * cout))
        w[f'conv{j}.kernel'] = rng.uniform(-lim, lim, (kh, kw, cin, cout)).astype(dtype)
        w[f'conv{j}.bias'] = np.zeros(cout, dtype)
        w[f'ln{j}.gamma'] = np.ones(g['out'], dtype)
        w[f'ln{j}.beta'] = np.zeros(g['out'], dtype)
        if randomize_affine:
            w[f'conv{j}.bias'] = rng.normal(0, 0.1, cout).astype(dtype)
            w[f'ln{j}.gamma'] = (1 + 0.2 * rng.normal(size=g['out'])).astype(dtype)
            w[f'ln{j}.beta'] = (0.1 * rng.normal(size=g['out'])).astype(dtype)
    flat = int(np.prod(geo[-1]['out']))
    sl = flat // emb_sz
    lim1 = np.sqrt(6.0 / (sl + 32))
    lim2 = np.sqrt(6.0 / (32 + 1))
    w['div.w1'] = rng.uniform(-lim1, lim1, (emb_sz, sl, 32)).astype(dtype)
    w['div.b1'] = np.zeros((emb_sz, 32), dtype)
    w['div.w2'] = rng.uniform(-lim2, lim2, (emb_sz, 32, 1)).astype(dtype)
    w['div.b2'] = np.zeros((emb_sz, 1), dtype)
    if randomize_affine:
        w['div.b1'] = rng.normal(0, 0.1, (emb_sz, 32)).astype(dtype)
        w['div.b2'] = rng.normal(0, 0.1, (emb_sz, 1)).astype(dtype)
    return w


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def layer_norm(x, gamma, beta, eps=LN_EPS):
    """keras LayerNormalization(axis=(1,2,3)) on (B,F,T,C)."""
    mu = x.mean(axis=(1, 2, 3), keepdims=True)
    var = ((x - mu) ** 2).mean(axis=(1, 2, 3), keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma[None] + beta[None]


def layer_norm1d(x, gamma_c, beta_c, eps=LN_EPS):
    """keras LayerNormalization(axis=-1) on (B,F,T,C) (nnfp.py:64-65)."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma_c + beta_c


def batch_norm_inference(x, gamma_c, beta_c, moving_mean, moving_var, eps=1e-3):
    """keras BatchNormalization(axis=-1) with training=False (nnfp.py:70-71; see the module docstring)."""
    return (x - moving_mean) / np.sqrt(moving_var + eps) * gamma_c + beta_c


def apply_norm(x, w, j, norm, dtype):
    g, b = w[f'ln{j}.gamma'].astype(dtype), w[f'ln{j}.beta'].astype(dtype)
    if norm == 'layer_norm2d':
        return layer_norm(x, g, b)
    if norm == 'layer_norm1d':
        return layer_norm1d(x, g, b)
    return batch_norm_inference(x, g, b, w[f'bn{j}.moving_mean'].astype(dtype), w[f'bn{j}.moving_variance'].astype(dtype))


def convert_norm(w, norm, seed=0, randomize=True):
    """A weight dict of `init_weights` with the normalisation tensors of another MODEL.BN: gamma / beta of shape (C,), and for
    batch normalisation the moving statistics `bn{j}.moving_mean` / `.moving_variance` (keras: 0 / 1; randomize: as a restored
    checkpoint could hold them)."""
    rng = np.random.default_rng(seed)
    w = dict(w)
    for j in range(16):
        C = w[f'conv{j}.bias'].shape[0]
        dt = w[f'conv{j}.bias'].dtype
        w[f'ln{j}.gamma'] = (1 + 0.2 * rng.normal(size=C)).astype(dt) if randomize else np.ones(C, dt)
        w[f'ln{j}.beta'] = (0.1 * rng.normal(size=C)).astype(dt) if randomize else np.zeros(C, dt)
        if norm not in ('layer_norm1d', 'layer_norm2d'):
            w[f'bn{j}.moving_mean'] = (0.2 * rng.normal(size=C)).astype(dt) if randomize else np.zeros(C, dt)
            w[f'bn{j}.moving_variance'] = rng.uniform(0.5, 2.0, C).astype(dt) if randomize else np.ones(C, dt)
    return w


def conv_same(x, kernel, bias, axis, stride, pad):
    """Dense 3-tap conv along one axis with TF SAME padding.

    x (B,F,T,Cin); kernel (kh,kw,Cin,Cout) with the 3 along T (axis='T', kh=1)
    or along F (axis='F', kw=1); stride = (sF,sT); pad=(before,after) on the
    3-tap axis.  The 1-tap axis is subsampled with its stride (k=1 needs no pad).
    """
    sF, sT = stride
    if axis == 'T':
        k3 = kernel[0]                      # (3,Cin,Cout)
        x = x[:, ::sF]
        xp = np.pad(x, ((0, 0), (0, 0), pad, (0, 0)))
        n_out = (xp.shape[2] - 3) // sT + 1
        out = sum(xp[:, :, k:k + sT * (n_out - 1) + 1:sT] @ k3[k] for k in range(3))
    else:
        k3 = kernel[:, 0]
        x = x[:, :, ::sT]
        xp = np.pad(x, ((0, 0), pad, (0, 0), (0, 0)))
        n_out = (xp.shape[1] - 3) // sF + 1
        out = sum(xp[:, k:k + sF * (n_out - 1) + 1:sF] @ k3[k] for k in range(3))
    return out + bias


def front_conv(feat, w, dtype=np.float64, taps=None, norm='layer_norm2d'):
    """FingerPrinter.front_conv (nnfp.py:210-218): (B,F,T,1) -> (B, F'*T'*C) flattened.

    `taps` (optional list) collects the per-conv LN outputs for stage-by-stage parity.  `norm`: MODEL.BN (nnfp.py:63-71).
    """
    x = np.asarray(feat, dtype=dtype)
    geo = conv_geometry(x.shape[1:])
    for j, g in enumerate(geo):
        x = conv_same(x, w[f'conv{j}.kernel'].astype(dtype), w[f'conv{j}.bias'].astype(dtype),
                      g['axis'], g['stride'], g['pad'])
        x = elu(x)
        x = apply_norm(x, w, j, norm, dtype)
        if taps is not None:
            taps.append(x)
    return x.reshape(x.shape[0], -1)


def div_enc(x, w, dtype=np.float64):
    """DivEncLayer.call (nnfp.py:141-156): (B,D) -> (B,Q); slice q = x[:, q*S:(q+1)*S]."""
    x = np.asarray(x, dtype=dtype)
    Q, S, H = w['div.w1'].shape
    xs = x.reshape(x.shape[0], Q, S)
    h = elu(np.einsum('bqs,qsh->bqh', xs, w['div.w1'].astype(dtype)) + w['div.b1'].astype(dtype)[None])
    y = np.einsum('bqh,qho->bqo', h, w['div.w2'].astype(dtype)) + w['div.b2'].astype(dtype)[None]
    return y[..., 0]


def l2_normalize(x, eps=1e-12):
    """tf.math.l2_normalize(axis=1) (nnfp.py:229)."""
    ss = (x * x).sum(axis=1, keepdims=True)
    return x / np.sqrt(np.maximum(ss, eps))


def fingerprinter(feat, w, dtype=np.float64, norm='layer_norm2d'):
    """FingerPrinter.call (nnfp.py:223-231): (B,256,32,1) -> (B,128) unit-norm."""
    return l2_normalize(div_enc(front_conv(feat, w, dtype, norm=norm), w, dtype))
