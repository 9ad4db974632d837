"""Layer tables of the oracle.  Test infrastructure.

``unet20_layers`` is the 10-encoder / 10-decoder table inlined in
cRFUNet.py:192-218; with c_in=2 and a 1-channel last decoder it is also the
"UNet default config" (UNet.py itself ships no table: its constructor takes
caller-built block lists, UNet.py:7-13).  ``dsunet_architecture`` is the
real-valued form of UDSS.set_size (UDSS.py:582-683) expressed as the dict that
DSUNet.py:507-525 consumes.
"""

# (kernel, stride, padding) per encoder, cRFUNet.py:193-204
_ENC_GEOM = [((7, 1), (1, 1), (3, 0)), ((1, 7), (1, 1), (0, 3)),
             ((7, 5), (2, 2), (3, 2)), ((7, 5), (2, 1), (3, 2)),
             ((5, 3), (2, 2), (2, 1)), ((5, 3), (2, 1), (2, 1)),
             ((5, 3), (2, 2), (2, 1)), ((5, 3), (2, 1), (2, 1)),
             ((5, 3), (2, 2), (2, 1)), ((5, 3), (2, 1), (2, 1))]

# (kernel, stride, padding, output_padding) per decoder, cRFUNet.py:207-218
_DEC_GEOM = [((5, 3), (2, 1), (2, 1), (0, 0)), ((5, 3), (2, 2), (2, 1), (0, 1)),
             ((5, 3), (2, 1), (2, 1), (0, 0)), ((5, 3), (2, 2), (2, 1), (0, 1)),
             ((5, 3), (2, 1), (2, 1), (0, 0)), ((5, 3), (2, 2), (2, 1), (0, 0)),
             ((7, 5), (2, 1), (3, 2), (0, 0)), ((7, 5), (2, 2), (3, 2), (0, 0)),
             ((1, 7), (1, 1), (0, 3), (0, 0)), ((7, 1), (1, 1), (3, 0), (0, 0))]


def unet20_layers(c_in=14, c_last=64, f_dim=32):
    """Rows of ctor arguments for UNet_m.Encoder / UNet_m.Decoder."""
    w = 2 * f_dim
    enc_ch = [(c_in, f_dim), (f_dim, w)] + [(w, w)] * 8
    dec_ch = [(w, w)] + [(2 * w, w)] * 8 + [(w + f_dim, c_last)]
    enc = [dict(args=(ci, co, k, s, p), kw={}) for (ci, co), (k, s, p) in zip(enc_ch, _ENC_GEOM)]
    dec = [dict(args=(ci, co, k, s), kw=dict(padding=p, output_padding=op))
           for (ci, co), (k, s, p, op) in zip(dec_ch, _DEC_GEOM)]
    return enc, dec


def default_unet(c_in=2, c_last=1, f_dim=32, blocks=None):
    """(encoders, decoders) lists for ``UNet(encoders, decoders)`` with the table
    above.  ``blocks`` = (EncoderClass, DecoderClass); defaults to the oracle's."""
    if blocks is None:
        from .blocks import Encoder, Decoder
        blocks = (Encoder, Decoder)
    enc_rows, dec_rows = unet20_layers(c_in=c_in, c_last=c_last, f_dim=f_dim)
    return ([blocks[0](*r["args"], **r["kw"]) for r in enc_rows],
            [blocks[1](*r["args"], **r["kw"]) for r in dec_rows])


def dsunet_architecture(model_complexity=45, input_channels=8):
    """Architecture dict for DSUNet: encoder channels
    [in, mc, mc, 2mc x7, 128], decoder i takes dec_channels[i] + the matching
    skip's channels and emits 2mc; output_padding (0,1) on decoders 1,3,5,7
    (UDSS.py:583-683).  Decoder run-order i is stored under key dec{10-i}."""
    mc = model_complexity
    enc_channels = [input_channels, mc, mc] + [2 * mc] * 7 + [128]
    dec_channels = [0] + [2 * mc] * 11
    dec_opad = [(0, 0), (0, 1)] * 4 + [(0, 0), (0, 0)]
    L = 10
    enc = {}
    for i in range(L):
        k, s, p = _ENC_GEOM[i]
        enc["enc{}".format(i + 1)] = dict(in_channels=enc_channels[i],
                                          out_channels=enc_channels[i + 1],
                                          kernel_size=k, stride=s, padding=p)
    dec = {}
    for i in range(L):
        k, s, p, _ = _DEC_GEOM[i]
        dec["dec{}".format(L - i)] = dict(in_channels=dec_channels[i] + enc_channels[L - i],
                                          out_channels=dec_channels[i + 1],
                                          kernel_size=k, stride=s, padding=p,
                                          output_padding=dec_opad[i])
    return dict(encoder=enc, decoder=dec)
