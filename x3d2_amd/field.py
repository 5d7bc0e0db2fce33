"""field_t and allocator_t mirrors
(/root/reference/src/field.f90:5-23, src/allocator.f90:64-179): a pool of
equally sized device blocks handed out with a direction tag and a data_loc.

Blocks are torch CUDA tensors (PyTorch is the device-memory plumbing); the
kernels only ever see their raw pointers."""
import os

import torch

from . import _lib
from .common import DIR_C, DIR_X, DIR_Y, DIR_Z, NULL_LOC, X3dError


class Field:
    """field_t: flat device block + dir + data_loc + refcount."""
    __slots__ = ("data", "dir", "data_loc", "refcount", "id", "owner")

    def __init__(self, data, ident, owner=None):
        self.data, self.dir, self.data_loc, self.refcount, self.id = data, DIR_X, NULL_LOC, 0, ident
        self.owner = owner

    @property
    def ptr(self):
        return self.data.data_ptr()

    def set_data_loc(self, data_loc):
        self.data_loc = data_loc

    def fill(self, c):
        """field_t%fill, src/field.f90:47-55"""
        lz = self.owner.lazy if self.owner is not None else None
        if lz is not None:  # deferred execution: the block address is a handle, only the library may write to it
            lib, h = lz
            if lib.x3d_block_fill(h, self.ptr, float(c)) != 0:
                raise X3dError(lib.x3d_last_error().decode())
            return
        self.data.fill_(c)


class Allocator:
    """allocator_t: LIFO free list, blocks are never freed before destroy
    (src/allocator.f90:113-179)."""

    def __init__(self, nblock_elems, device):
        self.n = int(nblock_elems)
        self.device = device
        self.free = []
        self.next_id = 0
        self.lazy = None  # (lib, backend handle) while the library's deferred execution is on (HipBackend(lazy=True))
        self._registered = []  # tensors whose memory the deferred-execution layer knows as blocks (kept alive)
        self.stagger = int(os.environ.get("X3D_BLOCK_STAGGER", str(self.STAGGER)))  # (read once)

    # Blocks start 4224 B (one padded row of a 512^3 block) further into their allocation than the previous one,
    # modulo 16: the kernels stream 4-12 blocks at the same relative offset at once, and with every block on a
    # 2 MiB boundary those streams hit the same memory channels at the same time (same-box A/B at 512^3:
    # 48.7 -> 47.7 ms per step, transeq_x 0.84 -> 0.79 ms per component; steps of 4 KB .. 270 KB give the same).
    STAGGER = 528

    def create_block(self):
        self.next_id += 1
        st = self.stagger
        if st:
            off = (self.next_id % 16) * st
            f = Field(torch.zeros(self.n + 16 * st, dtype=_lib.torch_real(), device=self.device)[off:off + self.n],
                      self.next_id, self)
        else:
            f = Field(torch.zeros(self.n, dtype=_lib.torch_real(), device=self.device), self.next_id, self)
        if self.lazy is not None:
            lib, h = self.lazy
            if lib.x3d_lazy_register_block(h, f.ptr) != 0:
                raise X3dError(lib.x3d_last_error().decode())
            # the layer treats a registered block's memory as reusable storage: the tensor must stay alive until the
            # block is unregistered (destroy), whatever happens to the Field object
            self._registered.append(f.data)
        return f

    def get_block(self, direction, data_loc=None):
        if direction not in (DIR_X, DIR_Y, DIR_Z, DIR_C):
            raise X3dError("Undefined direction, allocator cannot provide a shape.")
        f = self.free.pop() if self.free else self.create_block()
        f.dir = direction
        f.data_loc = NULL_LOC if data_loc is None else data_loc  # :139-145
        f.refcount = 1
        return f

    def release_block(self, f):
        f.refcount = 0
        self.free.append(f)
        if self.lazy is not None:  # the contents are dead until the block is written again
            lib, h = self.lazy
            if lib.x3d_block_discard(h, f.ptr) != 0:
                raise X3dError(lib.x3d_last_error().decode())

    def get_block_ids(self):
        return [f.id for f in reversed(self.free)]

    def destroy(self):
        self.free.clear()
        if self.lazy is not None:  # every handle's data home, then the layer forgets the blocks torch owns
            lib, h = self.lazy
            for t in self._registered:
                if lib.x3d_lazy_unregister_block(h, t.data_ptr()) != 0:
                    raise X3dError(lib.x3d_last_error().decode())
        self._registered.clear()
