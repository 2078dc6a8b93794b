"""Host-side mirror of the reference's `Rotation` / `Rigid` value types (mdgen/rigid_utils.py:282-1391)
for the subset used on the sampler path.  All arithmetic runs in
the fp32 HIP kernels of libmdgen_amd.so (csrc/k_se3.hip); PyTorch only carries device memory.
There is no CPU arithmetic path: operations on CPU tensors raise.

Field names `_rots._rot_mats`, `_rots._quats`, `_trans` are kept because the reference's drivers read
them directly (sim_inference.py:55-56, 92-93).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from ._lib import lib, launch, ptr, require_cuda


def _bcast(shape_a, shape_b):
    return torch.broadcast_shapes(tuple(shape_a), tuple(shape_b))


def _f32c(t):
    return t.to(torch.float32).contiguous()


class Rotation:
    """rigid_utils.py:282-810.  Exactly one of rot_mats [*,3,3] / quats [*,4] (w,x,y,z); fp32 forced."""

    def __init__(self, rot_mats: Optional[torch.Tensor] = None, quats: Optional[torch.Tensor] = None,
                 normalize_quats: bool = True):
        if (rot_mats is None) == (quats is None):
            raise ValueError("Exactly one input argument must be specified")
        if (rot_mats is not None and rot_mats.shape[-2:] != (3, 3)) or (quats is not None and quats.shape[-1] != 4):
            raise ValueError("Incorrectly shaped rotation matrix or quaternion")
        self._rot_mats = None if rot_mats is None else rot_mats.to(torch.float32)
        # quats / |quats| (rigid_utils.py:324-325) is applied inside the kernel when converting
        self._normalize = bool(normalize_quats) and quats is not None
        self._quats = None if quats is None else quats.to(torch.float32)

    # -- views ------------------------------------------------------------------------------------
    @property
    def shape(self):
        return self._rot_mats.shape[:-2] if self._rot_mats is not None else self._quats.shape[:-1]

    @property
    def device(self):
        return (self._rot_mats if self._rot_mats is not None else self._quats).device

    def __getitem__(self, index):
        if type(index) != tuple:
            index = (index,)
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats[index + (slice(None), slice(None))])
        r = Rotation(quats=self._quats[index + (slice(None),)], normalize_quats=False)
        r._normalize = self._normalize
        return r

    def unsqueeze(self, dim):
        if dim >= len(self.shape):
            raise ValueError("Invalid dimension")
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats.unsqueeze(dim if dim >= 0 else dim - 2))
        r = Rotation(quats=self._quats.unsqueeze(dim if dim >= 0 else dim - 1), normalize_quats=False)
        r._normalize = self._normalize
        return r

    @staticmethod
    def cat(rs: Sequence["Rotation"], dim: int) -> "Rotation":
        mats = [r.get_rot_mats() for r in rs]
        return Rotation(rot_mats=torch.cat(mats, dim=dim if dim >= 0 else dim - 2))

    def to(self, device=None, dtype=None):
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats.to(device=device))
        r = Rotation(quats=self._quats.to(device=device), normalize_quats=False)
        r._normalize = self._normalize
        return r

    def cuda(self):
        return self.to("cuda")

    # -- arithmetic (HIP) -------------------------------------------------------------------------
    def get_rot_mats(self) -> torch.Tensor:
        if self._rot_mats is not None:
            return self._rot_mats
        q = _f32c(self._quats)
        require_cuda(q)
        out = torch.empty(q.shape[:-1] + (3, 3), dtype=torch.float32, device=q.device)
        launch(lib.mdgen_quat_to_rot, q, q.numel() // 4, ptr(q), int(self._normalize), ptr(out))
        return out

    def get_quats(self) -> torch.Tensor:
        """rot_to_quat (rigid_utils.py:191-210); sign canonicalised to w >= 0 (the reference's eigh sign
        is arbitrary and is fixed afterwards by the caller, wrapper.py:309)."""
        if self._quats is not None:
            q = self._quats
            return q / torch.linalg.norm(q, dim=-1, keepdim=True) if self._normalize else q
        r = _f32c(self._rot_mats)
        require_cuda(r)
        out = torch.empty(r.shape[:-2] + (4,), dtype=torch.float32, device=r.device)
        launch(lib.mdgen_rot_to_quat, r, r.numel() // 9, ptr(r), ptr(out))
        return out

    def invert(self) -> "Rotation":
        return Rotation(rot_mats=self.get_rot_mats().transpose(-1, -2))

    def compose_r(self, r: "Rotation") -> "Rotation":
        z = torch.zeros(self.shape + (3,), dtype=torch.float32, device=self.device)
        z2 = torch.zeros(r.shape + (3,), dtype=torch.float32, device=self.device)
        return Rigid(self, z).compose(Rigid(r, z2)).get_rots()

    def apply(self, pts: torch.Tensor) -> torch.Tensor:
        return Rigid(self, torch.zeros(self.shape + (3,), dtype=torch.float32, device=self.device)).apply(pts)

    def invert_apply(self, pts: torch.Tensor) -> torch.Tensor:
        return Rigid(self, torch.zeros(self.shape + (3,), dtype=torch.float32, device=self.device)).invert_apply(pts)


class Rigid:
    """rigid_utils.py:813-1391: a rotation + translation per element of a virtual batch shape."""

    def __init__(self, rots: Optional[Rotation], trans: Optional[torch.Tensor]):
        if rots is None and trans is None:
            raise ValueError("At least one of rots / trans is required")
        if rots is None:
            eye = torch.eye(3, dtype=torch.float32, device=trans.device)
            rots = Rotation(rot_mats=eye.expand(trans.shape[:-1] + (3, 3)))
        if trans is None:
            trans = torch.zeros(rots.shape + (3,), dtype=torch.float32, device=rots.device)
        if rots.shape != trans.shape[:-1] or rots.device != trans.device:
            raise ValueError("Rots and trans incompatible")
        self._rots = rots
        self._trans = trans.to(torch.float32)

    # -- views ------------------------------------------------------------------------------------
    @staticmethod
    def identity(shape, dtype=None, device=None, requires_grad=False, fmt="rot_mat") -> "Rigid":
        eye = torch.eye(3, dtype=torch.float32, device=device).expand(tuple(shape) + (3, 3)).contiguous()
        return Rigid(Rotation(rot_mats=eye), torch.zeros(tuple(shape) + (3,), dtype=torch.float32, device=device))

    @property
    def shape(self):
        return self._trans.shape[:-1]

    @property
    def device(self):
        return self._trans.device

    def __getitem__(self, index):
        if type(index) != tuple:
            index = (index,)
        return Rigid(self._rots[index], self._trans[index + (slice(None),)])

    def unsqueeze(self, dim):
        if dim >= len(self.shape):
            raise ValueError("Invalid dimension")
        return Rigid(self._rots.unsqueeze(dim), self._trans.unsqueeze(dim if dim >= 0 else dim - 1))

    @staticmethod
    def cat(ts: Sequence["Rigid"], dim: int) -> "Rigid":
        return Rigid(Rotation.cat([t._rots for t in ts], dim),
                     torch.cat([t._trans for t in ts], dim=dim if dim >= 0 else dim - 1))

    def get_rots(self) -> Rotation:
        return self._rots

    def get_trans(self) -> torch.Tensor:
        return self._trans

    def to(self, device=None, dtype=None):
        return Rigid(self._rots.to(device=device), self._trans.to(device=device))

    def cuda(self):
        return self.to("cuda")

    def __mul__(self, right: torch.Tensor) -> "Rigid":
        """Pointwise multiply of all 9+3 entries by a mask (rigid_utils.py:923-942)."""
        return Rigid(Rotation(rot_mats=self._rots.get_rot_mats() * right[..., None, None]),
                     self._trans * right[..., None])

    __rmul__ = __mul__

    # -- arithmetic (HIP) -------------------------------------------------------------------------
    def _flat(self, shape):
        r = _f32c(self._rots.get_rot_mats().expand(tuple(shape) + (3, 3)))
        t = _f32c(self._trans.expand(tuple(shape) + (3,)))
        require_cuda(r, t)
        return r, t

    def compose(self, r: "Rigid") -> "Rigid":
        """rigid_utils.py:1031-1045."""
        shp = _bcast(self.shape, r.shape)
        r1, t1 = self._flat(shp)
        r2, t2 = r._flat(shp)
        ro, to = torch.empty_like(r1), torch.empty_like(t1)
        launch(lib.mdgen_rigid_compose, t1, t1.numel() // 3, ptr(r1), ptr(t1), ptr(r2), ptr(t2), ptr(ro), ptr(to))
        return Rigid(Rotation(rot_mats=ro), to)

    def invert(self) -> "Rigid":
        """rigid_utils.py:1075-1085."""
        r1, t1 = self._flat(self.shape)
        ro, to = torch.empty_like(r1), torch.empty_like(t1)
        launch(lib.mdgen_rigid_invert, t1, t1.numel() // 3, ptr(r1), ptr(t1), ptr(ro), ptr(to))
        return Rigid(Rotation(rot_mats=ro), to)

    def _apply(self, pts, inverse):
        shp = _bcast(self.shape, pts.shape[:-1])
        r1, t1 = self._flat(shp)
        p = _f32c(pts.expand(tuple(shp) + (3,)))
        require_cuda(p)
        out = torch.empty_like(p)
        launch(lib.mdgen_rigid_apply, p, p.numel() // 3, 1, ptr(r1), ptr(t1), ptr(p), ptr(out), int(inverse))
        return out

    def apply(self, pts: torch.Tensor) -> torch.Tensor:
        """rigid_utils.py:1047-1059."""
        return self._apply(pts, False)

    def invert_apply(self, pts: torch.Tensor) -> torch.Tensor:
        """rigid_utils.py:1061-1073."""
        return self._apply(pts, True)

    def to_tensor_7(self) -> torch.Tensor:
        """rigid_utils.py:1143-1155: [quat | trans] (quaternion sign w >= 0)."""
        return torch.cat([self._rots.get_quats(), self._trans], dim=-1)

    @staticmethod
    def from_tensor_7(t: torch.Tensor, normalize_quats: bool = False) -> "Rigid":
        """rigid_utils.py:1157-1173."""
        if t.shape[-1] != 7:
            raise ValueError("Incorrectly shaped input tensor")
        return Rigid(Rotation(quats=t[..., :4], normalize_quats=normalize_quats), t[..., 4:])

    @staticmethod
    def from_3_points(p_neg_x: torch.Tensor, origin: torch.Tensor, p_xy: torch.Tensor, eps: float = 1e-8) -> "Rigid":
        """rigid_utils.py:1175-1218: Gram-Schmidt frame from three points (`mdgen_from_3_points`)."""
        if eps != 1e-8:
            raise NotImplementedError("from_3_points is built for the reference's eps = 1e-8")
        a, o, b = torch.broadcast_tensors(p_neg_x, origin, p_xy)
        a, o, b = _f32c(a), _f32c(o), _f32c(b)
        require_cuda(a, o, b)
        rot = torch.empty(o.shape[:-1] + (3, 3), dtype=torch.float32, device=o.device)
        trans = torch.empty_like(o)
        launch(lib.mdgen_from_3_points, o, o.numel() // 3, ptr(a), ptr(o), ptr(b), ptr(rot), ptr(trans))
        return Rigid(Rotation(rot_mats=rot), trans)

    def map_tensor_fn(self, fn) -> "Rigid":
        """rigid_utils.py:1087-1107: apply a tensor -> tensor function to every rotation entry and translation
        coordinate (the leading "virtual" dims are what `fn` sees), e.g. `lambda x: torch.sum(x, dim=-1)` over a
        one-hot-masked group axis.  View-level glue: the function itself runs as the torch ops it is made of."""
        r = self._rots.get_rot_mats()
        new_r = torch.stack([fn(x) for x in torch.unbind(r.reshape(r.shape[:-2] + (9,)), dim=-1)], dim=-1)
        new_r = new_r.reshape(new_r.shape[:-1] + (3, 3))
        new_t = torch.stack([fn(x) for x in torch.unbind(self._trans, dim=-1)], dim=-1)
        return Rigid(Rotation(rot_mats=new_r), new_t)

    @staticmethod
    def from_tensor_4x4(t: torch.Tensor) -> "Rigid":
        """rigid_utils.py:1122-1141."""
        if t.shape[-2:] != (4, 4):
            raise ValueError("Incorrectly shaped input tensor")
        return Rigid(Rotation(rot_mats=t[..., :3, :3]), t[..., :3, 3])
