"""Topology of the detector forward graphs as plain node lists.

variant "A" (default): what the reference's ``UNINA_YOLO_DLA.forward`` computes (reference:
unina_yolo_dla/model.py:308-365, blocks at model.py:23-147, backbone :152-219, neck :224-269, head :274-303).
variant "B": the reference's QAT model ``UNINA_YOLO_DLA_QAT`` (unina_yolo_dla/qat.py:350-491) -- same blocks, but a
stride-32 stage (stage4_conv + SPPF at 16x base channels, qat.py:391-392), a third FPN level (qat.py:396-403), the PAN's
last concat on the FUSED p4 (qat.py:474) and flat module names (``stem``, ``stage1_c3k2``, ``head_p2_cls.0`` ...): the
layout its checkpoints (qat.py state_dicts) come in. Same output contract.
Nothing here executes arithmetic: the node list is consumed by

* ``synth.py``   -- to enumerate parameter names / shapes (state_dict keys match
                    the reference's, e.g. ``backbone.stem.conv.weight``),
* ``export.py``  -- to lower the graph to the engine's fused op table.

Node kinds
    conv      ConvBlock = Conv2d(bias=False,pad=k//2) -> BN(eval) -> ReLU   (model.py:23-50)
    convout   plain Conv2d 1x1 with bias, no BN, no activation            (model.py:292,299)
    add       residual add (model.py:73)
    cat       channel concat, order as listed
    pool5     MaxPool2d(5, stride 1, pad 2)                                 (model.py:125)
    up2       nearest-neighbour x2                                          (model.py:145-147)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class Node:
    kind: str
    name: str                 # module path (conv/convout) or synthetic name
    srcs: List[int]           # indices of producer nodes (-1 = network input)
    c: int                    # output channels
    h: int
    w: int
    k: int = 1
    s: int = 1
    cin: int = 0
    idx: int = -1
    users: List[int] = field(default_factory=list)


class Graph:
    """Builder + container. ``nodes[i]`` is in topological (forward) order."""

    def __init__(self, num_classes: int = 4, base_channels: int = 32, lite_p2: bool = False,
                 in_h: int = 640, in_w: int = 640, variant: str = "A"):
        if variant not in ("A", "B"):
            raise ValueError("variant must be 'A' (model.py) or 'B' (qat.py)")
        m = 16 if variant == "A" else 32
        if in_h % m or in_w % m:
            raise ValueError(f"input H and W must be multiples of {m} (stride-2 stages + x2 upsample/concat)")
        if variant == "B" and lite_p2:
            raise ValueError("lite_p2 exists only in graph (A) (model.py:184-190)")
        self.num_classes = num_classes
        self.base_channels = base_channels
        self.lite_p2 = lite_p2
        self.variant = variant
        self.in_h, self.in_w = in_h, in_w
        self.nodes: List[Node] = []
        self.outputs: List[int] = []          # p2_cls, p2_reg, p3_cls, p3_reg, p4_cls, p4_reg
        if variant == "A":
            self._build()
        else:
            self._build_b()

    # -- primitive emitters -------------------------------------------------
    def _add(self, n: Node) -> int:
        n.idx = len(self.nodes)
        self.nodes.append(n)
        for s in n.srcs:
            if s >= 0:
                self.nodes[s].users.append(n.idx)
        return n.idx

    def _shape(self, i: int):
        if i < 0:
            return 3, self.in_h, self.in_w
        n = self.nodes[i]
        return n.c, n.h, n.w

    def conv(self, name: str, src: int, cout: int, k: int = 3, s: int = 1) -> int:
        cin, h, w = self._shape(src)
        p = k // 2
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        return self._add(Node("conv", name, [src], cout, ho, wo, k, s, cin))

    def convout(self, name: str, src: int, cout: int) -> int:
        cin, h, w = self._shape(src)
        return self._add(Node("convout", name, [src], cout, h, w, 1, 1, cin))

    def add(self, name: str, a: int, b: int) -> int:
        c, h, w = self._shape(a)
        assert self._shape(b) == (c, h, w)
        return self._add(Node("add", name, [a, b], c, h, w))

    def cat(self, name: str, srcs: List[int]) -> int:
        shapes = [self._shape(i) for i in srcs]
        assert len({(h, w) for _, h, w in shapes}) == 1, shapes
        return self._add(Node("cat", name, list(srcs), sum(c for c, _, _ in shapes), shapes[0][1], shapes[0][2]))

    def pool5(self, name: str, src: int) -> int:
        c, h, w = self._shape(src)
        return self._add(Node("pool5", name, [src], c, h, w, 5, 1))

    def up2(self, name: str, src: int) -> int:
        c, h, w = self._shape(src)
        return self._add(Node("up2", name, [src], c, 2 * h, 2 * w))

    # -- composite blocks (reference: model.py:53-132) -----------------------
    def bottleneck(self, name: str, x: int, c: int) -> int:
        # Inside C3k2 always expansion=1.0, shortcut=True, in==out  (model.py:99)
        t = self.conv(f"{name}.cv1", x, c, k=1)
        t = self.conv(f"{name}.cv2", t, c, k=3)
        return self.add(f"{name}.add", x, t)

    def c3k2(self, name: str, x: int, cout: int, n: int) -> int:
        hid = int(cout * 0.5)
        p1 = self.conv(f"{name}.cv1", x, hid, k=1)
        p2 = self.conv(f"{name}.cv2", x, hid, k=1)
        for i in range(n):
            p1 = self.bottleneck(f"{name}.bottlenecks.{i}", p1, hid)
        cat = self.cat(f"{name}.cat", [p1, p2])          # order [path1, path2]  (model.py:110)
        return self.conv(f"{name}.cv3", cat, cout, k=1)

    def sppf(self, name: str, x: int, cout: int) -> int:
        cin, _, _ = self._shape(x)
        hid = cin // 2
        x = self.conv(f"{name}.cv1", x, hid, k=1)
        y1 = self.pool5(f"{name}.pool1", x)
        y2 = self.pool5(f"{name}.pool2", y1)
        y3 = self.pool5(f"{name}.pool3", y2)
        cat = self.cat(f"{name}.cat", [x, y1, y2, y3])   # model.py:132
        return self.conv(f"{name}.cv2", cat, cout, k=1)

    def head(self, name: str, x: int):
        c, _, _ = self._shape(x)
        outs = []
        for br, nout in (("cls", self.num_classes), ("reg", 4)):
            # graph (A): DetectionHead.cls_branch / .reg_branch (model.py:289-299); graph (B): head_pN_cls / head_pN_reg
            # Sequentials (qat.py:411-440)
            prefix = f"{name}.{br}_branch" if self.variant == "A" else f"{name}_{br}"
            t = self.conv(f"{prefix}.0", x, c, k=3)
            t = self.conv(f"{prefix}.1", t, c, k=3)
            outs.append(self.convout(f"{prefix}.2", t, nout))
        return outs

    # -- the network (model.py:205-219, 252-269, 357-365) --------------------
    def _build(self):
        bc = self.base_channels
        c1, c2, c3, c4 = bc, bc * 2, bc * 4, bc * 8
        x = self.conv("backbone.stem", -1, c1, k=3, s=2)
        x = self.conv("backbone.stage1_conv", x, c2, k=3, s=2)
        if self.lite_p2:
            p2 = self.conv("backbone.stage1_block", x, c2, k=3)
        else:
            p2 = self.c3k2("backbone.stage1_block", x, c2, n=1)
        x = self.conv("backbone.stage2_conv", p2, c3, k=3, s=2)
        p3 = self.c3k2("backbone.stage2_c3k2", x, c3, n=2)
        x = self.conv("backbone.stage3_conv", p3, c4, k=3, s=2)
        p4 = self.c3k2("backbone.stage3_c3k2", x, c4, n=2)
        p4_sppf = self.sppf("backbone.sppf", p4, c4)

        lat3 = self.conv("neck.lateral_p3", p4_sppf, c3, k=1)
        p4_up = self.up2("neck.up1", lat3)
        p3_fused = self.c3k2("neck.fpn_c3k2_1", self.cat("neck.cat_fpn1", [p4_up, p3]), c3, n=1)
        lat2 = self.conv("neck.lateral_p2", p3_fused, c2, k=1)
        p3_up = self.up2("neck.up2", lat2)
        p2_fused = self.c3k2("neck.fpn_c3k2_2", self.cat("neck.cat_fpn2", [p3_up, p2]), c2, n=1)
        p2_down = self.conv("neck.down1", p2_fused, c2, k=3, s=2)
        p3_out = self.c3k2("neck.pan_c3k2_1", self.cat("neck.cat_pan1", [p2_down, p3_fused]), c3, n=1)
        p3_down = self.conv("neck.down2", p3_out, c3, k=3, s=2)
        # NOTE: the last concat takes the PRE-SPPF p4 (model.py:254,267)
        p4_out = self.c3k2("neck.pan_c3k2_2", self.cat("neck.cat_pan2", [p3_down, p4]), c4, n=1)

        for hname, feat in (("head_p2", p2_fused), ("head_p3", p3_out), ("head_p4", p4_out)):
            self.outputs += self.head(hname, feat)

    # -- graph (B): UNINA_YOLO_DLA_QAT (qat.py:381-491) ------------------------
    def _build_b(self):
        bc = self.base_channels
        c1, c2, c3, c4, c5 = bc, bc * 2, bc * 4, bc * 8, bc * 16
        x = self.conv("stem", -1, c1, k=3, s=2)
        x = self.conv("stage1_conv", x, c2, k=3, s=2)
        p2 = self.c3k2("stage1_c3k2", x, c2, n=1)
        x = self.conv("stage2_conv", p2, c3, k=3, s=2)
        p3 = self.c3k2("stage2_c3k2", x, c3, n=2)
        x = self.conv("stage3_conv", p3, c4, k=3, s=2)
        p4 = self.c3k2("stage3_c3k2", x, c4, n=2)
        x = self.conv("stage4_conv", p4, c5, k=3, s=2)
        p5_sppf = self.sppf("stage4_sppf", x, c5)

        p5_up = self.up2("up_p5", self.conv("lateral_p4", p5_sppf, c4, k=1))
        p4_fused = self.c3k2("fpn_c3k2_1", self.cat("cat_fpn1", [p5_up, p4]), c4, n=1)        # qat.py:464
        p4_up = self.up2("up_p4", self.conv("lateral_p3", p4_fused, c3, k=1))
        p3_fused = self.c3k2("fpn_c3k2_2", self.cat("cat_fpn2", [p4_up, p3]), c3, n=1)        # qat.py:467
        p3_up = self.up2("up_p3", self.conv("lateral_p2", p3_fused, c2, k=1))
        p2_fused = self.c3k2("fpn_c3k2_3", self.cat("cat_fpn3", [p3_up, p2]), c2, n=1)        # qat.py:470
        p2_down = self.conv("down1", p2_fused, c2, k=3, s=2)
        p3_out = self.c3k2("pan_c3k2_1", self.cat("cat_pan1", [p2_down, p3_fused]), c3, n=1)  # qat.py:473
        p3_down = self.conv("down2", p3_out, c3, k=3, s=2)
        p4_out = self.c3k2("pan_c3k2_2", self.cat("cat_pan2", [p3_down, p4_fused]), c4, n=1)  # qat.py:476 (FUSED p4)
        for hname, feat in (("head_p2", p2_fused), ("head_p3", p3_out), ("head_p4", p4_out)):
            self.outputs += self.head(hname, feat)

    # -- derived facts ------------------------------------------------------
    def convs(self) -> List[Node]:
        return [n for n in self.nodes if n.kind in ("conv", "convout")]

    def macs(self) -> int:
        return sum(n.h * n.w * n.c * n.cin * n.k * n.k for n in self.convs())

    def param_shapes(self):
        """state_dict key -> shape, in the reference's key order per module."""
        out = {}
        for n in self.convs():
            if n.kind == "conv":
                out[f"{n.name}.conv.weight"] = (n.c, n.cin, n.k, n.k)
                for p in ("weight", "bias", "running_mean", "running_var"):
                    out[f"{n.name}.bn.{p}"] = (n.c,)
            else:
                out[f"{n.name}.weight"] = (n.c, n.cin, 1, 1)
                out[f"{n.name}.bias"] = (n.c,)
        return out


OUTPUT_NAMES = ("p2_cls", "p2_reg", "p3_cls", "p3_reg", "p4_cls", "p4_reg")   # model.py:383
STRIDES = (4, 8, 16)
