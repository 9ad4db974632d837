"""Layer tables used by the device modules.

The 10-encoder / 10-decoder geometry is the one cRFUNet.py:192-218 inlines
(kernel, stride, padding[, output_padding] per layer); the DSUNet dictionary
follows UDSS.set_size (UDSS.py:582-683) in the key layout DSUNet.py:507-525
reads.
"""

ENCODER_GEOMETRY = (
    # kernel, stride, padding
    ((7, 1), (1, 1), (3, 0)),
    ((1, 7), (1, 1), (0, 3)),
    ((7, 5), (2, 2), (3, 2)),
    ((7, 5), (2, 1), (3, 2)),
    ((5, 3), (2, 2), (2, 1)),
    ((5, 3), (2, 1), (2, 1)),
    ((5, 3), (2, 2), (2, 1)),
    ((5, 3), (2, 1), (2, 1)),
    ((5, 3), (2, 2), (2, 1)),
    ((5, 3), (2, 1), (2, 1)),
)

DECODER_GEOMETRY = (
    # kernel, stride, padding, output_padding
    ((5, 3), (2, 1), (2, 1), (0, 0)),
    ((5, 3), (2, 2), (2, 1), (0, 1)),
    ((5, 3), (2, 1), (2, 1), (0, 0)),
    ((5, 3), (2, 2), (2, 1), (0, 1)),
    ((5, 3), (2, 1), (2, 1), (0, 0)),
    ((5, 3), (2, 2), (2, 1), (0, 0)),
    ((7, 5), (2, 1), (3, 2), (0, 0)),
    ((7, 5), (2, 2), (3, 2), (0, 0)),
    ((1, 7), (1, 1), (0, 3), (0, 0)),
    ((7, 1), (1, 1), (3, 0), (0, 0)),
)


def unet20_layers(c_in=14, c_last=64, f_dim=32):
    """Constructor argument rows for UNet_m.Encoder / UNet_m.Decoder."""
    wide = 2 * f_dim
    enc, dec = [], []
    cin = c_in
    for idx, (k, s, p) in enumerate(ENCODER_GEOMETRY):
        cout = f_dim if idx == 0 else wide
        enc.append(dict(args=(cin, cout, k, s, p), kw={}))
        cin = cout
    for idx, (k, s, p, op) in enumerate(DECODER_GEOMETRY):
        if idx == 0:
            cin = wide                      # bottleneck output only
        elif idx == len(DECODER_GEOMETRY) - 1:
            cin = wide + f_dim              # skip from encoder 0
        else:
            cin = 2 * wide
        cout = c_last if idx == len(DECODER_GEOMETRY) - 1 else wide
        dec.append(dict(args=(cin, cout, k, s), kw=dict(padding=p, output_padding=op)))
    return enc, dec


def dsunet_architecture(model_complexity=45, input_channels=8):
    """``architecture`` dict for DSUNet: run-order decoder i lives under key
    ``dec{10-i}``; output_padding (0,1) on run-order decoders 1, 3, 5, 7."""
    mc = model_complexity
    enc_ch = [input_channels, mc, mc] + [2 * mc] * 7 + [128]
    n = len(ENCODER_GEOMETRY)
    arch = dict(encoder={}, decoder={})
    for i, (k, s, p) in enumerate(ENCODER_GEOMETRY):
        arch["encoder"]["enc{}".format(i + 1)] = dict(
            in_channels=enc_ch[i], out_channels=enc_ch[i + 1], kernel_size=k, stride=s, padding=p)
    for i, (k, s, p, _) in enumerate(DECODER_GEOMETRY):
        prev = 0 if i == 0 else 2 * mc
        arch["decoder"]["dec{}".format(n - i)] = dict(
            in_channels=prev + enc_ch[n - i], out_channels=2 * mc, kernel_size=k, stride=s,
            padding=p, output_padding=(0, 1) if i in (1, 3, 5, 7) else (0, 0))
    return arch
