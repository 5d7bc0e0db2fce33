"""Neighbour exchange = the reference's sendrecv_fields
(/root/reference/src/backend/omp/sendrecv.f90:10-36,
src/backend/cuda/sendrecv.f90:13-100) and the pencil transposes of the FFT
Poisson solver, on torch.distributed.

One process per GPU; backend "nccl" is RCCL over xGMI on ROCm.  Every pattern
is point-to-point with known peers (the two ring neighbours of a pencil
direction; the py or pz peers of a transpose), batched into one group
(batch_isend_irecv = ncclGroupStart/End) -- no bulk collective on the data path.
With the "gloo" backend (CPU tests, or several ranks sharing one GPU in the
GPU parity tests) device buffers are staged through host memory."""
import os

import torch
import torch.distributed as dist


class _Done:
    """an exchange that has already completed (self-exchange, host-staged transport)"""

    def wait(self):
        pass


DONE = _Done()


class _Pending:
    """an exchange in flight on RCCL's stream: wait() makes the CURRENT stream wait for it (no host block)"""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []


class _PendingEvent:
    """an exchange that is complete when `event` (recorded on the communication stream behind it) is"""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class Comm:
    # verdict of the overlapped path's self-check per (group, backend, self-via-RCCL): the check is collective and
    # allocates 512 MiB -- every Comm object of a process after the first reuses the verdict (all ranks construct their
    # Comm objects in the same order, so they agree on who checks); Comm(self_check=False) opts out (ordered path)
    _verdicts = {}

    def __init__(self, group=None, self_check=True):
        self.enabled = dist.is_available() and dist.is_initialized()
        self.group = group
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.size = dist.get_world_size(group) if self.enabled else 1
        self.backend = dist.get_backend(group) if self.enabled else None
        self.host_staged = self.backend == "gloo"
        # exchanges started on a second HIP stream overlap with kernels of the compute stream that do not
        # depend on them (X3D_NO_OVERLAP=1: every exchange is ordered on the compute stream, for A/B runs)
        self.overlap = os.environ.get("X3D_NO_OVERLAP") != "1"
        self._cstream = None
        self._cycles_per_s = None
        self.stream_probe = None  # ms of each candidate's probe (_pick_stream), None until a stream was needed
        # the overlapped path is verified against the ordered one on first use (self_check below): a transport
        # whose stream semantics differ from what _start assumes costs the overlap, not the results
        # X3D_COMM_SELF_VIA_NCCL=1 (one GPU, world size 1, X3D_EMULATE_DECOMP): exchanges with THIS rank go through
        # RCCL send / recv to itself instead of device copies -- the only way to put the RCCL code path (group launch,
        # communication stream, wait semantics) under a test on a one-GPU box
        self.self_via_nccl = self.enabled and self.backend == "nccl" and os.environ.get("X3D_COMM_SELF_VIA_NCCL") == "1"
        # X3D_COMM_FAKE_PEERS=1 (world size 1 only): EVERY peer of an all-to-all is this rank -- chunk i of the send buffer
        # lands in slot i of the receive buffer, as a device copy or (with X3D_COMM_SELF_VIA_NCCL) as N send / recv pairs
        # to self in one RCCL group: the group sizes, chunk lengths and part pipelining of an N-rank all-to-all run through
        # RCCL on a one-GPU box (the numbers mean nothing: tests compare the two transports bit for bit)
        self.fake_peers = self.size == 1 and os.environ.get("X3D_COMM_FAKE_PEERS") == "1"
        # X3D_COMM_EMULATE_LINKS=<GB/s per link and direction> (world size 1, exchanges through RCCL to self): every
        # exchange additionally holds its stream for the time the message would spend on xGMI links of that rate -- an
        # all-to-all among n peers sends 1 / n of its bytes over each of n - 1 links at once (bytes / n / rate), a
        # neighbour exchange its two messages over two links at once (largest message / rate) -- so that a one-GPU box
        # shows which part of the transfers a schedule hides (bench.py --virtual-ranks; a timeline, not a measurement of links)
        self.link_rate = None
        e = os.environ.get("X3D_COMM_EMULATE_LINKS")
        if e and self.size == 1:
            self.link_rate = float(e) * 1e9
            # with the links emulated a message to self moves as a device copy on the communication stream (RCCL's
            # send / recv to self would add a second, real copy through its own kernels on top of the emulated link
            # time); every group additionally holds the stream X3D_COMM_EMULATE_LATENCY_US (default 30: what one RCCL
            # point-to-point group costs to launch on this pool, profiles/r05_*) -- the RCCL calls themselves stay under
            # test in the bit-for-bit RCCL-to-self tests, which run without the emulation
            self.link_latency = float(os.environ.get("X3D_COMM_EMULATE_LATENCY_US", "30")) * 1e-6
        self._checked = self.host_staged or not self.enabled or not self.overlap or \
            (self.size == 1 and not self.self_via_nccl)
        self.self_check_result = None
        self.self_check_error = None
        # event timers around the exchanges (bench.py switches them on for ONE step outside its timed region):
        # kind -> [count, bytes, [(start event, end event)]]
        self.timed = False
        self._timing = {}
        # the check is collective (a ring exchange + an all-reduce): it runs HERE, where every rank is -- not at the first
        # exchange, which a rank whose first group is empty would skip while its neighbours wait for it
        if not self._checked:
            # (keyed on WHO is in the group, not on id(group): CPython reuses ids after garbage collection and a new group
            #  may have other members or another transport -- ADVICE round 5)
            members = None
            if group is not None:
                try:
                    members = tuple(dist.get_process_group_ranks(group))
                except Exception:
                    members = ("group", id(group), self.size)
            key = (members, self.size, self.backend, self.self_via_nccl)
            if not self_check:
                self._checked, self.overlap = True, False
            elif key in Comm._verdicts:
                self._checked = True
                self.self_check_result, self.self_check_error = Comm._verdicts[key]
                self.overlap = self.overlap and bool(self.self_check_result)
            else:
                self.self_check()
                Comm._verdicts[key] = (self.self_check_result, getattr(self, "self_check_error", None))

    # ------------------------------------------------------------ p2p core
    def _note(self, kind, sends, e0, e1):
        t = self._timing.setdefault(kind, [0, 0, []])
        t[0] += 1
        t[1] += sum(x.numel() * x.element_size() for x, _ in sends)
        t[2].append((e0, e1))

    def timing_report(self):
        """{kind: {"exchanges", "MB_sent", "ms"}} of the exchanges timed since `timed` was switched on (HIP events on the
        stream the transfers were posted from: for the overlapped path the time from "the producers are done" to "RCCL's
        work has completed", which includes RCCL's own queueing)"""
        import torch as _t
        _t.cuda.synchronize()
        out = {}
        for kind, (n, nbytes, evs) in self._timing.items():
            out[kind] = {"exchanges": n, "MB_sent": nbytes / 1e6, "ms": float(sum(a.elapsed_time(b) for a, b in evs))}
        self._timing = {}
        return out

    def _exchange(self, sends, recvs, kind="sendrecv"):
        """sends: [(tensor, peer)], recvs: [(tensor, peer)]; posting order
        defines the matching between a pair of ranks"""
        if not sends and not recvs:
            return
        if self.timed:  # (host-staged dry runs too: the events then bracket the host's copies, i.e. wall time)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.timed = False
            try:
                self._exchange(sends, recvs, kind)
            finally:
                self.timed = True
            e1.record()
            self._note(kind, sends, e0, e1)
            return
        if self.host_staged:
            s_host = [(t.detach().to("cpu", copy=True), p) for t, p in sends]
            r_host = [(torch.empty(t.shape, dtype=t.dtype, device="cpu"), p) for t, p in recvs]
            ops = [dist.P2POp(dist.isend, t, p, self.group) for t, p in s_host]
            ops += [dist.P2POp(dist.irecv, t, p, self.group) for t, p in r_host]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for (dst, _), (src, _) in zip(recvs, r_host):
                dst.copy_(src)
            return
        if self.link_rate and all(p == self.rank for _, p in sends):
            for (src, _), (dst, _) in zip(sends, recvs):  # (see _start_overlapped)
                n = min(src.numel(), dst.numel())
                dst[:n].copy_(src[:n])
            self._link_hold(sends, kind)
            return
        ops = [dist.P2POp(dist.isend, t, p, self.group) for t, p in sends]
        ops += [dist.P2POp(dist.irecv, t, p, self.group) for t, p in recvs]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if self.link_rate:
            self._link_hold(sends, kind)

    def _comm_stream(self):
        if self._cstream is None:
            self._cstream = self._pick_stream()
        return self._cstream

    def _sleep_cycles_per_s(self):
        if getattr(self, "_cycles_per_s", None) is None:  # calibrate torch.cuda._sleep's unit once
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(1000)
            torch.cuda.synchronize()
            e0.record()
            torch.cuda._sleep(20_000_000)
            e1.record()
            torch.cuda.synchronize()
            self._cycles_per_s = 20_000_000 / (e0.elapsed_time(e1) * 1e-3)
        return self._cycles_per_s

    def _pick_stream(self):
        """a stream whose work really runs BESIDE the current stream's.  HIP maps streams onto a few hardware queues
        (GPU_MAX_HW_QUEUES, 4 by default) round robin, and two streams that share a queue run strictly one after the
        other: measured on the pool's boxes, one torch pool stream in about six shares the compute stream's queue and
        hides NOTHING of what is posted on it (scratch/overlap_probe2.py, profiles/r05_stream_queues.txt; with it the
        whole overlapped exchange path ran at the ordered path's speed).  So candidates are probed -- a 1 ms one-wave
        spin on the candidate against one on the current stream: side by side they take 1 ms, on one queue 2 -- and a
        colliding one is kept referenced (the pool then hands out the next) and passed over."""
        import time
        self.stream_probe = []
        if os.environ.get("X3D_COMM_NO_STREAM_PROBE") == "1":
            return torch.cuda.Stream()
        n = int(2e-3 * self._sleep_cycles_per_s())

        def two_spins(side):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if side is None:
                torch.cuda._sleep(n)
            else:
                with torch.cuda.stream(side):
                    torch.cuda._sleep(n)
            torch.cuda._sleep(n)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3
        serial = two_spins(None)  # both on the current stream: what sharing a queue costs
        self._colliding = []
        s = None
        for _ in range(6):  # (few candidates: every stream in use takes a hardware queue, and too many of them are time-sliced)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):  # a stream's first launch sets its queue up: not part of the measurement
                torch.cuda._sleep(1000)
            ms = two_spins(s)
            self.stream_probe.append((round(ms, 2), round(serial, 2)))
            if ms < 0.75 * serial:
                return s
            self._colliding.append(s)
        return s

    def _link_hold(self, sends, kind):
        """emulated link time of this exchange, spent on the current stream (X3D_COMM_EMULATE_LINKS)"""
        cps = self._sleep_cycles_per_s()
        sizes = [t.numel() * t.element_size() for t, _ in sends]
        if not sizes:
            return
        if kind == "alltoall":
            n = len(sizes)
            sec = (sum(sizes) / n) / self.link_rate if n > 1 else 0.0  # (n chunks incl. the one a rank keeps)
        else:
            half = len(sizes) // 2 or 1  # [(to prev, to next)] x pairs: the two directions use two links at once
            sec = max(sum(sizes[0::2]), sum(sizes[1::2])) / self.link_rate if half else 0.0
        sec += self.link_latency
        # the message to self has just moved as a device copy on this stream; on a node the link transfer IS that copy
        # (RCCL's kernels read and write HBM while the link carries the bytes): the two are not additive -- the hold is
        # what the link adds beyond the copy, priced at 2.5 TB/s of payload (5 TB/s of read + write traffic, what the
        # streaming kernels of this backend reach; its HBM traffic beside the compute kernels stays real)
        sec -= sum(sizes) / 2.5e12
        if sec > 0.0:
            torch.cuda._sleep(int(sec * cps))

    def _start(self, sends, recvs, kind="sendrecv"):
        """post a group of point-to-point transfers behind everything queued on the current stream, on the
        communication stream: kernels launched on the compute stream afterwards run beside it"""
        if not sends and not recvs:
            return DONE
        if self.host_staged or not self.overlap:
            self._exchange(sends, recvs, kind)
            return DONE
        return self._start_overlapped(sends, recvs, kind)

    def _start_overlapped(self, sends, recvs, kind="sendrecv"):
        cs = self._comm_stream()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):  # RCCL's own stream waits for the stream that is current at posting time
            if self.timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.link_rate and all(p == self.rank for _, p in sends):
                for (src, _), (dst, _) in zip(sends, recvs):  # (posting order pairs them: same peer, same position)
                    n = min(src.numel(), dst.numel())  # (uneven shares of the pencil solver: a virtual peer's chunk
                    dst[:n].copy_(src[:n])             #  need not have this rank's length; timing only)
                works = []
            else:
                ops = [dist.P2POp(dist.isend, t, p, self.group) for t, p in sends]
                ops += [dist.P2POp(dist.irecv, t, p, self.group) for t, p in recvs]
                works = dist.batch_isend_irecv(ops)
            if self.timed or self.link_rate:  # (the communication stream waits for RCCL's work; the compute stream still waits by itself)
                for w in works:
                    w.wait()
                if self.link_rate:
                    self._link_hold(sends, kind)
                if self.timed:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self._note(kind, sends, e0, e1)
            if self.link_rate:
                done = torch.cuda.Event()
                done.record()
                return _PendingEvent(done)
        return _Pending(works)

    def self_check(self):
        """ONE ring exchange through the overlapped path (communication stream, handle.wait()) against the same
        exchange through the ordered path, with a producer kernel queued right before the send and a consumer right
        after the wait -- the two orderings _start relies on (RCCL's work waits for the stream that is current when it
        is posted; wait() makes the compute stream wait for RCCL's).  All ranks agree on the verdict (all-reduce);
        a mismatch or an error switches the overlap off for this run (results stay right, exchanges are ordered on
        the compute stream) and says so on rank 0.  X3D_NO_OVERLAP=1 skips the check and the overlap."""
        self._checked = True
        ok = 1.0
        try:
            n = 1 << 21
            dev = torch.device("cuda", torch.cuda.current_device())
            prev, nxt = (self.rank - 1) % self.size, (self.rank + 1) % self.size
            heavy = torch.ones(1 << 26, dtype=torch.float64, device=dev)  # (512 MiB, freed below)
            out = []
            for overlapped in (False, True):
                send = torch.zeros(n, dtype=torch.float64, device=dev)
                recv = torch.full((n,), -1.0, dtype=torch.float64, device=dev)
                # producer: a long kernel, then the values to send, both on the compute stream
                heavy.mul_(1.0000001)
                send.copy_(torch.arange(n, dtype=torch.float64, device=dev) + 1000.0 * (self.rank + 1))
                sends, recvs = [(send, nxt)], [(recv, prev)]
                if overlapped:
                    h = self._start_overlapped(sends, recvs)
                    heavy.mul_(1.0000001)  # (independent work beside the transfer)
                    h.wait()
                else:
                    self._exchange(sends, recvs)
                out.append(recv.clone())  # consumer on the compute stream
            want = torch.arange(n, dtype=torch.float64, device=dev) + 1000.0 * (prev + 1)
            torch.cuda.synchronize()
            if not (torch.equal(out[0], want) and torch.equal(out[1], want)):
                ok = 0.0
            del heavy, out, want
        except Exception as e:  # noqa: BLE001 -- whatever it is, the ordered path is the fallback
            ok = 0.0
            self.self_check_error = repr(e)
        try:
            t = torch.tensor([ok], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            ok = float(t.item())
        except Exception:  # noqa: BLE001
            ok = 0.0
        self.self_check_result = bool(ok)
        if not ok:
            self.overlap = False
            if self.rank == 0:
                print("x3d2_amd.parallel: overlapped exchange failed its self-check against the ordered path; "
                      "exchanges are ordered on the compute stream for this run", flush=True)

    # ------------------------------------------------------------ halo / boundary exchange
    def sendrecv(self, pairs, prev, nxt):
        """pairs: list of (send_s, send_e, recv_s, recv_e) tensors.
        send_s -> prev (arrives in prev's recv_e), send_e -> next (arrives in
        next's recv_s)."""
        if self.fake_peers:
            prev = nxt = self.rank
        if prev == self.rank and nxt == self.rank and not self.self_via_nccl:
            for send_s, send_e, recv_s, recv_e in pairs:  # nproc == 1 branch, sendrecv.f90:20-22
                recv_s.copy_(send_e)
                recv_e.copy_(send_s)
            return
        sends, recvs = [], []
        for send_s, send_e, recv_s, recv_e in pairs:
            # with two ranks prev == nxt: the peer posts (send_s, send_e) in this
            # order too, so its send_s must land in our recv_e and its send_e in recv_s
            sends += [(send_s, prev), (send_e, nxt)]
            recvs += [(recv_e, nxt), (recv_s, prev)]
        self._exchange(sends, recvs)

    def isendrecv(self, pairs, prev, nxt):
        """sendrecv started now and completed by the returned handle's wait()"""
        if self.fake_peers:
            prev = nxt = self.rank
        if prev == self.rank and nxt == self.rank and not self.self_via_nccl:
            self.sendrecv(pairs, prev, nxt)
            return DONE
        sends, recvs = [], []
        for send_s, send_e, recv_s, recv_e in pairs:
            sends += [(send_s, prev), (send_e, nxt)]
            recvs += [(recv_e, nxt), (recv_s, prev)]
        return self._start(sends, recvs)

    # ------------------------------------------------------------ transposes
    def alltoall(self, sendbuf, send_counts, recvbuf, recv_counts, peers):
        """personalised exchange among `peers` (global ranks, own rank
        included at its position): chunk i of sendbuf goes to peers[i], chunk i of
        recvbuf comes from peers[i]; counts in elements of the buffers' dtype."""
        so = ro = 0
        sends, recvs = [], []
        for cnt_s, cnt_r, peer in zip(send_counts, recv_counts, peers):
            peer = self.rank if self.fake_peers else peer
            if peer == self.rank and not self.self_via_nccl:
                if not (recvbuf.data_ptr() == sendbuf.data_ptr() and ro == so):  # (aliased buffers: nothing to move)
                    recvbuf[ro:ro + cnt_r].copy_(sendbuf[so:so + cnt_s])
            else:
                if cnt_s:
                    sends.append((sendbuf[so:so + cnt_s], peer))
                if cnt_r:
                    recvs.append((recvbuf[ro:ro + cnt_r], peer))
            so += cnt_s
            ro += cnt_r
        self._exchange(sends, recvs, "alltoall")

    def ialltoall(self, sendbuf, recvbuf, count, peers, send_off=0, send_stride=None, recv_off=0, recv_stride=None):
        """one part of a personalised exchange, started now: peer i gets `count` elements that start at
        send_off + i * send_stride of sendbuf and delivers `count` elements to recv_off + i * recv_stride of
        recvbuf (strides default to count: packed).  The slab Poisson solver sends its spectrum in several such
        parts so that the z transforms of one part run beside the transfer of the others."""
        ss = count if send_stride is None else send_stride
        rs = count if recv_stride is None else recv_stride
        sends, recvs = [], []
        for i, peer in enumerate(peers):
            peer = self.rank if self.fake_peers else peer
            s0, r0 = send_off + i * ss, recv_off + i * rs
            if peer == self.rank and not self.self_via_nccl:
                if not (recvbuf.data_ptr() == sendbuf.data_ptr() and r0 == s0):
                    recvbuf[r0:r0 + count].copy_(sendbuf[s0:s0 + count])
            else:
                sends.append((sendbuf[s0:s0 + count], peer))
                recvs.append((recvbuf[r0:r0 + count], peer))
        return self._start(sends, recvs, "alltoall")

    def ialltoall_chunks(self, sendbuf, recvbuf, chunks, peers):
        """several ialltoall parts in ONE group (one RCCL group launch): chunks = [(count, off, stride)], peer i gets /
        delivers `count` elements at off + i * stride of both buffers for every chunk (the y-slab Poisson solver's rows
        groups: a rows group's pieces of the kz groups' blocks are separate contiguous runs per peer)"""
        sends, recvs = [], []
        for count, off, stride in chunks:
            for i, peer in enumerate(peers):
                peer = self.rank if self.fake_peers else peer
                o = off + i * stride
                if peer == self.rank and not self.self_via_nccl:
                    if recvbuf.data_ptr() != sendbuf.data_ptr():
                        recvbuf[o:o + count].copy_(sendbuf[o:o + count])
                else:
                    sends.append((sendbuf[o:o + count], peer))
                    recvs.append((recvbuf[o:o + count], peer))
        return self._start(sends, recvs, "alltoall")

    def ialltoallv(self, sendbuf, send_offs, send_counts, recvbuf, recv_offs, recv_counts, peers):
        """ialltoall with a chunk per peer of its own offset and length (the pencil Poisson solver's groups of
        planes: x modes / y rows are shared out unevenly)"""
        sends, recvs = [], []
        for s0, cs, r0, cr, peer in zip(send_offs, send_counts, recv_offs, recv_counts, peers):
            peer = self.rank if self.fake_peers else peer
            if peer == self.rank and not self.self_via_nccl:
                if cs != cr:
                    raise ValueError("ialltoallv: the chunk a rank keeps has one length")
                if cs and not (recvbuf.data_ptr() == sendbuf.data_ptr() and r0 == s0):
                    recvbuf[r0:r0 + cr].copy_(sendbuf[s0:s0 + cs])
            else:
                if cs:
                    sends.append((sendbuf[s0:s0 + cs], peer))
                if cr:
                    recvs.append((recvbuf[r0:r0 + cr], peer))
        return self._start(sends, recvs, "alltoall")

    # ------------------------------------------------------------ scalars
    def allreduce(self, value, op="sum"):
        if self.size == 1:
            return value
        t = torch.tensor([value], dtype=torch.float64)
        if self.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        if self.size > 1:
            dist.barrier(group=self.group)
