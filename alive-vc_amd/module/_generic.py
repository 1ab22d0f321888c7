"""Generic-width path of the two frame-rate encoders (ContentEncoder, F0Estimator) and the logits of F0Estimator.forward:
the reference's constructors take sizes (/root/reference/module/content_encoder.py:9-14, f0_estimator.py:9-14) and
F0Estimator.forward returns the class logits (f0_estimator.py:22-27).  The fused entry points (alive_content_encoder,
alive_f0_estimate) are built for the default architecture and never store the 16-KB-per-frame logits; here the same layers are
composed from the op-level entry points of the C ABI (alive_conv1d on the fp32-grade three-plane split or the exact f32 MFMA,
alive_dwconv_norm, alive_channel_norm), which take any width.  One launch per layer: a slow path by design, HIP all the way
(no torch arithmetic), used by non-default constructors and by `forward` only."""
from . import ops


def _precision(ci):
    # the split kernel wants whole 32-channel blocks of input; anything else goes through the exact f32-MFMA kernel
    return "bf16x6" if ci % 32 == 0 and ci >= 64 else "fp32"


def convnext_stack(sd, x, num_layers, prefix="mid_layers."):
    """ConvNeXt1d x num_layers (common.py:45-62): dw k7 -> ChannelNorm -> 1x1 -> GELU -> 1x1 -> * scale -> + res"""
    for i in range(num_layers):
        p = f"{prefix}{i}"
        y = ops.dwconv_norm(x, sd[p + ".dw_conv.weight"], sd[p + ".dw_conv.bias"], gain=sd[p + ".norm.scale"], offset=sd[p + ".norm.shift"])
        c, hdim = sd[p + ".pw_conv1.weight"].shape[1], sd[p + ".pw_conv1.weight"].shape[0]
        hid, _ = ops.conv1d(y, sd[p + ".pw_conv1.weight"], sd[p + ".pw_conv1.bias"], act="gelu", precision=_precision(c))
        x, _ = ops.conv1d(hid, sd[p + ".pw_conv2.weight"], sd[p + ".pw_conv2.bias"], ch_scale=sd[p + ".scale"].reshape(-1),
                          residual=x, precision=_precision(hdim))
    return x


def content_encoder(sd, spec, num_layers):
    """ContentEncoder.forward (content_encoder.py:22-25)"""
    x, _ = ops.conv1d(spec, sd["input_layer.weight"], sd["input_layer.bias"], precision="fp32")
    x = convnext_stack(sd, x, num_layers)
    y, _ = ops.conv1d(x, sd["output_layer.weight"], sd["output_layer.bias"], precision=_precision(x.shape[1]))
    return y


def f0_logits(sd, spec, num_layers):
    """F0Estimator.forward (f0_estimator.py:22-27): logits [N, classes, T]"""
    x, _ = ops.conv1d(spec, sd["input_layer.weight"], sd["input_layer.bias"], precision="fp32")
    x = convnext_stack(sd, x, num_layers)
    x = ops.channel_norm(x, sd["last_norm.scale"], sd["last_norm.shift"])
    y, _ = ops.conv1d(x, sd["output_layer.weight"], sd["output_layer.bias"], precision=_precision(x.shape[1]))
    return y
