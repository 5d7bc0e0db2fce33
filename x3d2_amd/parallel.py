"""Neighbour exchange = the reference's sendrecv_fields
(/root/reference/src/backend/omp/sendrecv.f90:10-36,
src/backend/cuda/sendrecv.f90:13-100) on torch.distributed.

One process per GPU; backend "nccl" is RCCL over xGMI on ROCm.  The pattern is
point-to-point with the two ring neighbours of the pencil direction, batched
into one group (batch_isend_irecv = ncclGroupStart/End), never a collective."""
import torch
import torch.distributed as dist


class Comm:
    def __init__(self, group=None):
        self.enabled = dist.is_available() and dist.is_initialized()
        self.group = group
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.size = dist.get_world_size(group) if self.enabled else 1

    def sendrecv(self, pairs, prev, nxt):
        """pairs: list of (send_s, send_e, recv_s, recv_e) tensors.
        send_s -> prev (arrives in prev's recv_e), send_e -> next
        (arrives in next's recv_s); tag-free ordering = posting order."""
        if prev == self.rank and nxt == self.rank:
            for send_s, send_e, recv_s, recv_e in pairs:  # nproc == 1 branch, :20-22
                recv_s.copy_(send_e)
                recv_e.copy_(send_s)
            return
        ops = []
        for send_s, send_e, recv_s, recv_e in pairs:
            ops.append(dist.P2POp(dist.isend, send_s, prev, self.group))
            ops.append(dist.P2POp(dist.irecv, recv_e, nxt, self.group))
            ops.append(dist.P2POp(dist.isend, send_e, nxt, self.group))
            ops.append(dist.P2POp(dist.irecv, recv_s, prev, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def allreduce(self, value, op="sum"):
        if self.size == 1:
            return value
        t = torch.tensor([value], dtype=torch.float64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        if self.size > 1:
            dist.barrier(group=self.group)
