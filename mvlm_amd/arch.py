"""Static description of the landmark-heatmap network (MVLMModel) used on the hot path.

The network is the two-stack hourglass regressor of the reference
(src/mvlm/prediction/paulsenpredictor.py:364-432, ResidualBlock :251-273,
HourGlassModule :276-361).  This module only enumerates *what exists*: the
state-dict keys with their shapes, and the canonical conv-slot order that the
weight packer (mvlm_amd/weights.py) and the C++ graph executor
(mvlm_amd/csrc/cnn_graph.hip) both walk.  Nothing here computes.
"""
from __future__ import annotations

from dataclasses import dataclass

N_FEATURES = 256
BN_EPS = 1e-5

# image_channels string -> number of network input channels
# (paulsenpredictor.py:371-383)
IMAGE_CHANNELS = {
    "geometry": 1,
    "RGB": 3,
    "depth": 1,
    "RGB+depth": 4,
    "geometry+depth": 2,
}

# Which planes of the renderer's [N,256,256,4] RGB+depth stack feed the network
# for each mode.  "geometry" is a build-defined shaded plane (absent from the
# reference renderer) stored by our renderer in plane 0 when
# requested; see DESIGN.md.
CHANNEL_SELECT = {
    "RGB": (0, 1, 2),
    "depth": (3,),
    "RGB+depth": (0, 1, 2, 3),
    "geometry": (0,),
    "geometry+depth": (0, 3),
}


@dataclass(frozen=True)
class RBSpec:
    """One pre-activation residual block (paulsenpredictor.py:251-273)."""

    name: str  # state-dict prefix, e.g. "hg1.rb7"
    cin: int
    cout: int

    @property
    def has_resample(self) -> bool:
        return self.cin != self.cout


def residual_blocks() -> list[RBSpec]:
    """The 43 residual blocks in canonical (state-dict) order."""
    f = N_FEATURES
    rbs = [
        RBSpec("conv2", f // 4, f // 2),
        RBSpec("conv3", f // 2, f // 2),
        RBSpec("conv4", f // 2, f),
    ]
    for hg in ("hg1", "hg2"):
        for i in range(1, 21):
            rbs.append(RBSpec(f"{hg}.rb{i}", f, f))
    return rbs


@dataclass(frozen=True)
class ConvSlot:
    """One convolution in canonical slot order.

    ``pre_bn``/``post_bn`` name the BatchNorm whose folded scale/shift is applied
    (with ReLU) to the conv's input / output.  ``present`` is False for the
    identity-resample slot of blocks with cin == cout.
    """

    index: int
    name: str  # state-dict prefix of the conv ("conv1", "hg1.rb3.conv2", "conv2.resample.2")
    cin: int
    cout: int
    ksize: int
    has_bias: bool
    pre_bn: str | None
    post_bn: str | None
    present: bool = True


def conv_slots(n_landmarks: int, in_channels: int) -> list[ConvSlot]:
    """Canonical conv order shared with the C++ executor.

    slot 0            conv1 (+bias, post bn1+relu)
    slots 1+4r..4+4r  residual block r: resample(1x1) / conv1 / conv2 / conv3
    then              conv5, conv6, conv7, conv9, conv10, conv11
    conv8 is dead at inference (only the last stack stage is consumed,
    paulsenpredictor.py:204-205) and is not packed.
    """
    f = N_FEATURES
    nl = n_landmarks
    slots: list[ConvSlot] = []

    def add(name, cin, cout, k, bias, pre, post, present=True):
        slots.append(ConvSlot(len(slots), name, cin, cout, k, bias, pre, post, present))

    add("conv1", in_channels, f // 4, 3, True, None, "bn1")
    for rb in residual_blocks():
        p = rb.name
        add(f"{p}.resample.2", rb.cin, rb.cout, 1, False, f"{p}.resample.0", None, rb.has_resample)
        add(f"{p}.conv1", rb.cin, rb.cout // 2, 3, False, f"{p}.bn1", None)
        add(f"{p}.conv2", rb.cout // 2, rb.cout // 4, 3, False, f"{p}.bn2", None)
        add(f"{p}.conv3", rb.cout // 4, rb.cout // 4, 3, False, f"{p}.bn3", None)
    add("conv5", f, f, 3, True, None, "bn2")
    add("conv6", f, nl, 3, True, None, None)
    add("conv7", nl, f, 3, True, None, None)
    add("conv9", f, f, 3, True, None, "bn3")
    add("conv10", f, nl, 3, True, None, None)
    add("conv11", nl, nl, 3, True, None, None)
    # conv11 runs on a nearest-2x-upsampled input (paulsenpredictor.py:428-429): every output
    # pixel of parity (a, b) only sees a 2x2 window of the low-resolution tensor, so the layer is
    # also packed as four 2x2 convolutions with pre-summed taps (weights.collapse_upsampled_3x3)
    for a in (0, 1):
        for b in (0, 1):
            add(f"conv11.parity{a}{b}", nl, nl, 2, True, None, None)
    return slots


N_CONV_SLOTS = 1 + 4 * 43 + 6 + 4  # 183


def state_dict_shapes(n_landmarks: int, in_channels: int) -> dict[str, tuple[int, ...]]:
    """All 817 state-dict keys of MVLMModel with their shapes
    (paulsenpredictor.py:385-402), conv8 included."""
    f = N_FEATURES
    nl = n_landmarks
    shapes: dict[str, tuple[int, ...]] = {}

    def bn(prefix, c):
        shapes[f"{prefix}.weight"] = (c,)
        shapes[f"{prefix}.bias"] = (c,)
        shapes[f"{prefix}.running_mean"] = (c,)
        shapes[f"{prefix}.running_var"] = (c,)
        shapes[f"{prefix}.num_batches_tracked"] = ()

    def conv(prefix, cin, cout, k, bias):
        shapes[f"{prefix}.weight"] = (cout, cin, k, k)
        if bias:
            shapes[f"{prefix}.bias"] = (cout,)

    conv("conv1", in_channels, f // 4, 3, True)
    bn("bn1", f // 4)
    for rb in residual_blocks():
        p = rb.name
        bn(f"{p}.bn1", rb.cin)
        conv(f"{p}.conv1", rb.cin, rb.cout // 2, 3, False)
        bn(f"{p}.bn2", rb.cout // 2)
        conv(f"{p}.conv2", rb.cout // 2, rb.cout // 4, 3, False)
        bn(f"{p}.bn3", rb.cout // 4)
        conv(f"{p}.conv3", rb.cout // 4, rb.cout // 4, 3, False)
        if rb.has_resample:
            bn(f"{p}.resample.0", rb.cin)
            conv(f"{p}.resample.2", rb.cin, rb.cout, 1, False)
    conv("conv5", f, f, 3, True)
    bn("bn2", f)
    conv("conv6", f, nl, 3, True)
    conv("conv7", nl, f, 3, True)
    conv("conv8", nl, nl, 3, True)
    conv("conv9", f, f, 3, True)
    bn("bn3", f)
    conv("conv10", f, nl, 3, True)
    conv("conv11", nl, nl, 3, True)
    return shapes


def live_conv_flops_per_view(n_landmarks: int, in_channels: int) -> float:
    """2*Cin*Cout*k^2*H*W summed over the 138 live convs."""
    sizes = conv_spatial_sizes()
    total = 0.0
    for s in conv_slots(n_landmarks, in_channels):
        if s.present and s.ksize != 2:  # the parity slots restate conv11, already counted
            hw = sizes[s.name]
            total += 2.0 * s.cin * s.cout * s.ksize * s.ksize * hw * hw
    return total


def conv_spatial_sizes() -> dict[str, int]:
    """Output side length of every packed conv (paulsenpredictor.py:301-361, :404-432)."""
    sizes = {"conv1": 256, "conv5": 128, "conv6": 128, "conv7": 128, "conv9": 128, "conv10": 128, "conv11": 256}
    rb_size = {"conv2": 256, "conv3": 128, "conv4": 128}
    hg_level = {1: 128, 2: 64, 3: 64, 4: 32, 5: 32, 6: 16, 7: 16, 8: 8, 9: 8, 10: 4, 11: 4, 12: 4,
                13: 8, 14: 8, 15: 16, 16: 16, 17: 32, 18: 32, 19: 64, 20: 64}
    for hg in ("hg1", "hg2"):
        for i, s in hg_level.items():
            rb_size[f"{hg}.rb{i}"] = s
    for p, s in rb_size.items():
        for c in ("resample.2", "conv1", "conv2", "conv3"):
            sizes[f"{p}.{c}"] = s
    for ab in ("00", "01", "10", "11"):
        sizes[f"conv11.parity{ab}"] = 128  # evaluated on the low-resolution grid
    return sizes
