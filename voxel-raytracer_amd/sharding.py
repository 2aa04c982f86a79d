"""Row sharding of one frame over the GPUs of a node and the gather back to rank 0.

Pixels are independent, so the frame is cut into tiles of `tile_rows` rows dealt round-robin to the
ranks (sky rows are several times cheaper than model rows; interleaving balances them). Each rank
traces its tiles into ONE compact device buffer of int32 words laid out as

    [ rgba8 block : rows_max * W ][ id/dist block : rows_max * 2W ]        (3 * rows_max "rows" of W words)

so the only exchange step of the path -- the gather of finished rows to rank 0 -- is a single
collective per frame (RCCL over xGMI with backend "nccl", gloo in the CPU tests). Rank 0 receives
straight into one [world, 3 * rows_max, W] tensor and un-interleaves it with ONE index_copy_ into a
[3H + 1, W] frame store whose first H rows are the rgba8 image and next 2H rows the (voxelID, dist)
image (row 3H swallows the padding rows of ranks that own fewer rows).
"""
import torch
import torch.distributed as dist


class ShardPlan:
    def __init__(self, width, height, tile_rows, rank, world):
        self.width, self.height, self.tile_rows, self.rank, self.world = width, height, tile_rows, rank, world
        tiles = (height + tile_rows - 1) // tile_rows
        self.rows_of = []
        for r in range(world):
            rows = []
            for t in range(r, tiles, world):
                rows.extend(range(t * tile_rows, min(height, (t + 1) * tile_rows)))
            self.rows_of.append(rows)
        self.rows_local = len(self.rows_of[rank])
        self.rows_max = max(len(r) for r in self.rows_of)
        self.words = self.rows_max * width * 3  # int32 words per rank buffer

    def local_buffer(self, device):
        return torch.zeros(self.words, dtype=torch.int32, device=device)

    def pointers(self, buf):
        """device addresses of the rgba block and the id/dist block inside a rank buffer"""
        base = buf.data_ptr()
        return base, base + self.rows_max * self.width * 4

    # ---- rank 0 side -------------------------------------------------------------------------
    def gather_buffer(self, device):
        """[world, words] receive tensor; row r is rank r's compact buffer"""
        return torch.zeros((self.world, self.words), dtype=torch.int32, device=device)

    def frame_store(self, device):
        """[3H + 1, W] int32: rows [0,H) rgba8, rows [H,3H) id/dist (two W-word rows per pixel row), row 3H scratch"""
        return torch.zeros((3 * self.height + 1, self.width), dtype=torch.int32, device=device)

    def frame_views(self, store):
        H, W = self.height, self.width
        return store[:H], store[H:3 * H].view(H, W, 2)

    def scatter_index(self, device):
        """destination row in the frame store of every W-word row of the gathered tensor"""
        H, rm = self.height, self.rows_max
        idx = torch.full((self.world, 3 * rm), 3 * H, dtype=torch.long)
        for r, rows in enumerate(self.rows_of):
            for j, y in enumerate(rows):
                idx[r, j] = y
                idx[r, rm + 2 * j] = H + 2 * y
                idx[r, rm + 2 * j + 1] = H + 2 * y + 1
        return idx.reshape(-1).to(device)


def gather_frame(plan, local, gathered, store, index, group=None, stage_through_host=False, collective=None):
    """Gathers every rank's compact buffer to rank 0 and scatters the rows into the frame store there.

    local: this rank's buffer [words]; gathered: rank 0's [world, words] tensor (None elsewhere);
    store/index: rank 0's frame store and scatter index (None elsewhere).
    stage_through_host: for backends without device-tensor gather (gloo rehearsals of the N>1 path on a
    box whose ranks share one GPU): the collective runs on host copies."""
    if plan.world > 1 if collective is None else collective:   # collective=True: call the backend even with one rank
        if stage_through_host and local.is_cuda:
            host = local.cpu()
            recv = [torch.empty_like(host) for _ in range(plan.world)] if plan.rank == 0 else None
            dist.gather(host, recv, dst=0, group=group)
            if plan.rank == 0:
                gathered.copy_(torch.stack(recv))
        else:
            dist.gather(local, list(gathered.unbind(0)) if plan.rank == 0 else None, dst=0, group=group)
        src = gathered
    else:
        src = local
    if plan.rank != 0:
        return
    if plan.world == 1:   # one rank: its buffer holds the rows in frame order, the scatter is the identity: a plain copy
        store[:3 * plan.height].copy_(src.view(3 * plan.height, plan.width))
    else:
        store.index_copy_(0, index, src.view(plan.world * 3 * plan.rows_max, plan.width))


class FramePipeline:
    """B = max(2, streams) shard buffers per rank; frame i is traced into buffer i % B (on stream i % streams).

    gather="final" (default): frames stay where they were traced -- rank r keeps rows_of[r] of every frame
    resident in its HBM, exactly as the one-GPU run keeps whole frames resident -- and only drain() moves data:
    it gathers the LAST frame to rank 0 and assembles it there. The per-pixel path has no exchange step, so the
    steady state has no collective at all; a consumer of the sharded frames (a sharded display pass, an encoder
    per rank) reads them in place.

    gather="frame": every frame is delivered to rank 0 (what a single display GPU needs). The gather of frame i
    is issued asynchronously (RCCL runs it on its own stream behind an event) and overlaps the trace of frame
    i + 1; the only waits are the ones data hazards need: before buffer i % B is overwritten by frame i + B,
    gather i must have completed -- at that point rank 0 also scatters frame i into the frame store. A step then
    costs max(trace, gather) once the pipe is full. 12 B/pixel into ONE GPU bounds this mode: at 1080p that is
    21.8 MB x (N-1)/N per frame through rank 0's xGMI links.
    """

    def __init__(self, plan, device, group=None, stage_through_host=False, gather="final", streams=1, collective=None):
        if gather not in ("final", "frame"):
            raise ValueError("gather must be 'final' or 'frame'")
        if streams not in (1, 2, 3, 4):
            raise ValueError("streams must be 1 to 4")
        self.plan, self.group, self.via_host, self.mode = plan, group, stage_through_host, gather
        # collective=True runs the gathers through the backend even when there is a single rank (tools/rccl_rehearsal.py)
        self.dist = plan.world > 1 if collective is None else bool(collective)
        self.n_buf = nb = max(2, streams)
        self.local = [plan.local_buffer(device) for _ in range(nb)]
        root = plan.rank == 0
        n_recv = nb if gather == "frame" else 1
        self.gathered = [plan.gather_buffer(device) if (root and self.dist) else None for _ in range(n_recv)]
        self.store = plan.frame_store(device) if root else None
        self.index = plan.scatter_index(device) if root else None
        self.pending = [None] * nb    # per slot: None | "resident" | async work handle
        self.in_place = None          # one rank: the slot whose buffer is the newest complete frame (see _retire)
        self.frame = 0
        # streams > 1: buffer k is traced on HIP stream k, so the drain of one launch (its last, slowest waves) overlaps
        # the ramp-up of the next ones; frames i and i + B share a buffer AND a stream, so they stay ordered. An eighth
        # of a 1080p frame takes 25.7 us per launch on one stream, 13.7 on two, 9.2 on four (tools/shard_rate.py).
        self.cuda = torch.device(device).type == "cuda"
        if self.cuda and streams > 1:
            self.streams = [torch.cuda.Stream(device) for _ in range(nb)]
            for st in self.streams:
                st.wait_stream(torch.cuda.current_stream(device))
        elif self.cuda:
            self.streams = [torch.cuda.current_stream(device)] * nb
        else:
            self.streams = [None] * nb
        self.side = self.cuda and streams > 1

    def stream_handle(self, k):
        """hipStream_t (as int) the frame in slot k must be traced on"""
        return self.streams[k].cuda_stream if self.cuda else None

    def slot(self):
        """(slot index, rgba pointer, id pointer) of the buffer the NEXT frame must be traced into; makes that
        buffer safe to overwrite first. Trace on stream_handle(slot index)."""
        k = self.frame % self.n_buf
        self._retire(k)
        p_rgba, p_id = self.plan.pointers(self.local[k])
        return k, p_rgba, p_id

    def submit(self, k):
        """call after the trace of the frame in slot k has been enqueued on stream_handle(k)"""
        plan = self.plan
        if not self.dist or self.mode == "final":
            self.pending[k] = "resident"
        elif self.via_host and self.local[k].is_cuda:
            self._join(k)
            gather_frame(plan, self.local[k], self.gathered[k], self.store, self.index, self.group, True, self.dist)
            self.pending[k] = None
        else:
            recv = list(self.gathered[k].unbind(0)) if plan.rank == 0 else None
            if self.side:
                with torch.cuda.stream(self.streams[k]):   # the collective orders itself after this stream's trace
                    self.pending[k] = dist.gather(self.local[k], recv, dst=0, group=self.group, async_op=True)
            else:
                self.pending[k] = dist.gather(self.local[k], recv, dst=0, group=self.group, async_op=True)
        self.frame += 1

    def _join(self, k):
        """the current stream waits for everything enqueued on slot k's stream"""
        if self.side:
            torch.cuda.current_stream().wait_stream(self.streams[k])

    def _release(self, k):
        """slot k's stream waits for what the current stream has enqueued (reads of the slot's buffers)"""
        if self.side:
            self.streams[k].wait_stream(torch.cuda.current_stream())

    def _retire(self, k, final=False):
        h = self.pending[k]
        if h is None:
            return
        plan = self.plan
        self.pending[k] = None
        if h == "resident":
            # the frame stays in its shard buffer(s); only the last one is assembled on rank 0, on drain
            if final and plan.world == 1 and not self.dist:
                # one rank: its buffer holds every row in frame order -- it IS the frame ([3H, W]: rgba rows, then id/dist
                # rows), there is nothing to assemble and nothing is copied
                self._join(k)
                self.in_place = k
                return
            if final:
                self._join(k)
                gather_frame(plan, self.local[k], self.gathered[0], self.store, self.index, self.group,
                             self.via_host, self.dist)
                self._release(k)
            return
        h.wait()
        if plan.rank == 0:
            self.store.index_copy_(0, self.index, self.gathered[k].view(plan.world * 3 * plan.rows_max, plan.width))
        self._release(k)

    def drain(self):
        """completes the frames still in flight, oldest first; afterwards rank 0's store holds the newest frame"""
        nb = self.n_buf
        newest = (self.frame - 1) % nb
        for i in range(nb):                        # slot of frame (self.frame - nb + i): oldest ... newest
            k = (self.frame + i) % nb
            self._retire(k, final=(k == newest) or self.mode == "frame")
        if self.side:
            for k in range(nb):
                self._join(k)

    def frame_views(self):
        if self.in_place is not None:
            return self.plan.frame_views(self.local[self.in_place].view(3 * self.plan.height, self.plan.width))
        return self.plan.frame_views(self.store)


class PeerFramePipeline:
    """Every frame delivered with NO collective: the ranks' trace kernels store their row tiles straight into a frame
    buffer that lives on the frame's root rank, through an IPC mapping of that buffer (xGMI peer stores; include/vrt.h
    vrt_dispatch_tiles / vrt_ipc_*). Ordering is by stream-ordered flags in the root's memory, no host in the loop. A root
    owns n_buf frame buffers ("slots"); the i-th frame of a root uses slot k = i % n_buf and, on every rank, HIP stream k
    (so the drain of one launch overlaps the start of the next ones, and a slot's frames stay ordered):

        arrived[root][r][k]  rank r's stream k writes i + 1 after its tiles of the root's i-th frame
        consumed[root][k]    the root's consumer stream writes i + 1 once it has seen every rank's `arrived` of frame i; a
                             rank waits for consumed[k] >= i - n_buf + 1 before it overwrites slot k with frame i

    rotate=False: rank 0 is the root of every frame (a single display head: all 12 B/pixel cross rank 0's links).
    rotate=True : frame f is assembled on rank f % world (a consumer per GPU -- encoder, display pass -- takes every
                  world-th frame): the inbound traffic spreads over all GPUs' links.
    The control plane (exchange of the IPC handles) runs over the process group once, at construction.
    """

    def __init__(self, ctx, plan, n_buf=4, rotate=False, group=None, colour_only=False):
        import numpy as np
        self.ctx, self.plan, self.n_buf, self.rotate, self.group = ctx, plan, n_buf, rotate, group
        # colour_only: only the rgba8 image (4 of the 12 bytes per pixel) crosses to the root; a rank's (voxelID, dist) rows
        # stay in its own memory -- where a display pass sharded the same way would read them. A third of the link traffic.
        self.colour_only = colour_only
        self.own_id = []
        W, H, world, rank = plan.width, plan.height, plan.world, plan.rank
        self.roots = list(range(world)) if rotate else [0]
        self.own = rank in self.roots
        self.np = np
        self.n_flags = (world + 1) * n_buf          # 64 bytes apart
        self.local = None
        self.maps = {}
        self.stuck = False            # a device-side wait that never came back: the process must not synchronise again
        self.streams, self.consumer = [], None
        # Every rank makes the SAME collective calls in the same order whatever fails locally (a rank that raised early
        # would leave the others inside a collective): failures travel as text and every rank raises after the exchange.
        mine, err = None, None
        if self.own:
            try:
                self.local = {"rgba": [], "id": [], "flags": None}
                for _ in range(n_buf):
                    self.local["rgba"].append(ctx.device_alloc(W * H * 4))
                    self.local["id"].append(ctx.device_alloc(W * H * 8))
                self.local["flags"] = ctx.device_alloc(64 * self.n_flags)
                mine = {"rgba": [ctx.ipc_export(p) for p in self.local["rgba"]], "id": [ctx.ipc_export(p) for p in self.local["id"]],
                        "flags": ctx.ipc_export(self.local["flags"])}
            except Exception as ex:  # noqa: BLE001
                err = f"rank {rank}: {type(ex).__name__}: {ex}"
        gathered = [None] * world
        if world > 1:
            dist.all_gather_object(gathered, (mine, err), group=group)
        else:
            gathered = [(mine, err)]
        err = next((e for _, e in gathered if e), None)
        if err is None:
            try:
                if colour_only:
                    for _ in range(n_buf):
                        self.own_id.append(ctx.device_alloc(W * H * 8))
                for root in self.roots:
                    if root == rank:
                        self.maps[root] = self.local
                    else:
                        h = gathered[root][0]
                        m = {"rgba": [], "id": [], "flags": None}
                        self.maps[root] = m          # registered first: close() unmaps whatever was opened
                        for x in h["rgba"]:
                            m["rgba"].append(ctx.ipc_open(x))
                        for x in h["id"]:
                            m["id"].append(ctx.ipc_open(x))
                        m["flags"] = ctx.ipc_open(h["flags"])
            except Exception as ex:  # noqa: BLE001
                err = f"rank {rank}: {type(ex).__name__}: {ex}"
        errs = [err]
        if world > 1:
            errs = [None] * world
            dist.all_gather_object(errs, err, group=group)
        err = next((e for e in errs if e), None)
        if err is not None:
            self.close()
            raise RuntimeError("peer frame buffers: " + err)
        self.frame = 0
        self.seen = {root: 0 for root in self.roots}   # frames of each root enqueued so far
        self.streams = [torch.cuda.Stream() for _ in range(n_buf)]
        self.consumer = torch.cuda.Stream() if self.own else None

    def _arrived(self, root, r, k):
        return self.maps[root]["flags"] + 64 * (r * self.n_buf + k)

    def _consumed(self, root, k):
        return self.maps[root]["flags"] + 64 * (self.plan.world * self.n_buf + k)

    def rehearse(self, timeout_s=10.0):
        """One flag hand-shake per (rank, root) pair with HOST-side polling only (nothing can hang): every rank writes
        an arrival flag through the mapping, every root checks that the value shows up in its own memory. Returns the
        reason as text when the mappings do not behave, else None."""
        import time
        world, rank = self.plan.world, self.plan.rank

        def everyone(err):
            """the first error text of any rank; every rank calls this at the same points whatever happened to it locally"""
            out = [err]
            if world > 1:
                out = [None] * world
                dist.all_gather_object(out, err, group=self.group)
            return next((e for e in out if e), None)

        def guarded(fn):
            try:
                return fn()
            except Exception as ex:  # noqa: BLE001 -- an unsupported call on one rank must not leave the others in a collective
                return f"rank {rank}: {type(ex).__name__}: {ex}"

        def write_arrivals():
            for root in self.roots:
                self.ctx.stream_write_flag(self._arrived(root, rank, 0), 0x7000 + rank, self.streams[0].cuda_stream)
            if not self._poll([self.streams[0]], timeout_s):
                self.stuck = True
                return f"rank {rank}: a flag write through the mapping did not complete within {timeout_s} s"
            return None

        bad = everyone(guarded(write_arrivals))      # also the barrier between the writes and the roots' reads
        if bad:
            return bad

        def read_arrivals():
            if not self.own:
                return None
            t0 = time.time()
            want = [0x7000 + r for r in range(world)]
            while True:
                got = [int(self.ctx.device_read(self.local["flags"] + 64 * r * self.n_buf, (1,), self.np.uint32)[0]) for r in range(world)]
                if got == want:
                    break
                if time.time() - t0 > timeout_s:
                    return f"rank {rank}: arrival flags {got} != {want} after {timeout_s} s"
                time.sleep(0.01)
            self.ctx.device_write(self.local["flags"], self.np.zeros(16 * self.n_flags, self.np.uint32))
            return None

        bad = everyone(guarded(read_arrivals))
        if bad:
            return bad
        # Second half: a DEVICE-side wait on a flag in a root's memory (what step() enqueues for slot reuse). Every rank's
        # stream 0 waits for consumed[root][0] >= 0x7100 of every root, the roots then write that value, and the host polls
        # the stream with a time limit. A wait that does not come back is first offered the value through this rank's own
        # mapping; if that does not release it either the pipeline is marked stuck (the caller must not synchronise it).
        def enqueue_waits():
            for root in self.roots:
                self.ctx.stream_wait_flag(self._consumed(root, 0), 0x7100, self.streams[0].cuda_stream)
            return None

        bad = everyone(guarded(enqueue_waits))       # every wait is enqueued before any root releases it

        def release_and_poll():
            if self.own:
                self.ctx.stream_write_flag(self._consumed(rank, 0), 0x7100, self.consumer.cuda_stream)
            if self._poll([self.streams[0]], timeout_s):
                return None
            why = f"rank {rank}: a stream wait on a root's flag did not return within {timeout_s} s"
            for root in self.roots:
                self.ctx.stream_write_flag(self._consumed(root, 0), 0x7100, self.streams[1].cuda_stream)
            if not self._poll([self.streams[0]], 2.0):
                self.stuck = True
                why += " (and stays blocked)"
            return why

        bad2 = everyone(guarded(release_and_poll))   # run even after a failed enqueue: waits that did get enqueued must be released
        bad = bad or bad2

        def reset_flags():
            if not bad and self.own:
                if not self._poll([self.consumer], timeout_s):
                    self.stuck = True
                    return f"rank {rank}: the consumer stream did not drain"
                self.ctx.device_write(self.local["flags"], self.np.zeros(16 * self.n_flags, self.np.uint32))
            return None

        bad3 = everyone(guarded(reset_flags))        # doubles as the closing barrier
        return bad or bad3

    @staticmethod
    def _poll(streams, timeout_s):
        """True once every stream has drained, False after timeout_s (host-side spinning: nothing here can block)"""
        import time
        t0 = time.time()
        while True:
            if all(st.query() for st in streams):
                return True
            if time.time() - t0 > timeout_s:
                return False

    def step(self, dispatch_tiles):
        """enqueue frame self.frame: dispatch_tiles(d_rgba, d_id, stream) must trace this rank's tiles on that stream"""
        f, world, rank = self.frame, self.plan.world, self.plan.rank
        root = f % world if self.rotate else 0
        i = self.seen[root]                  # index of this frame among the root's frames
        k = i % self.n_buf
        stream = self.streams[k].cuda_stream
        if i >= self.n_buf:                  # the slot's previous frame must have been consumed
            self.ctx.stream_wait_flag(self._consumed(root, k), i - self.n_buf + 1, stream)
        if self.plan.rows_local:
            d_id = self.own_id[k] if (self.colour_only and root != rank) else self.maps[root]["id"][k]
            dispatch_tiles(self.maps[root]["rgba"][k], d_id, stream)
        self.ctx.stream_write_flag(self._arrived(root, rank, k), i + 1, stream)
        if root == rank:                     # this rank's consumer: frame complete once every rank has arrived
            cs = self.consumer.cuda_stream
            for r in range(world):
                self.ctx.stream_wait_flag(self._arrived(root, r, k), i + 1, cs)
            self.ctx.stream_write_flag(self._consumed(root, k), i + 1, cs)
        self.seen[root] = i + 1
        self.frame += 1

    def drain(self, timeout_s=None):
        """waits for everything enqueued; with a time limit it spins on the streams instead and returns False (and marks the
        pipeline stuck) when they do not drain -- a flag hand-shake that never completes must not take the process with it"""
        if timeout_s is not None:
            ok = self._poll(self.streams + ([self.consumer] if self.consumer is not None else []), timeout_s)
            self.stuck = self.stuck or not ok
            return ok
        for st in self.streams:
            st.synchronize()
        if self.consumer is not None:
            self.consumer.synchronize()
        return True

    def last_frame(self):
        """(rgba [H, W, 4] uint8, id [H, W, 2] int32) of the newest frame assembled on THIS rank (None if it roots none)"""
        if not self.own or self.seen.get(self.plan.rank, 0) == 0:
            return None
        H, W = self.plan.height, self.plan.width
        k = (self.seen[self.plan.rank] - 1) % self.n_buf
        return (self.ctx.device_read(self.local["rgba"][k], (H, W, 4), self.np.uint8),
                self.ctx.device_read(self.local["id"][k], (H, W, 2), self.np.int32))

    def close(self):
        if self.stuck:     # freeing under a blocked stream would block too: the memory goes with the process
            self.maps, self.local, self.own_id = {}, None, []
            return
        for root, m in self.maps.items():
            if root != self.plan.rank and m is not self.local:
                for p in m["rgba"] + m["id"] + ([m["flags"]] if m["flags"] else []):
                    self.ctx.ipc_close(p)
        if self.local:
            for p in self.local["rgba"] + self.local["id"] + ([self.local["flags"]] if self.local["flags"] else []):
                self.ctx.device_free(p)
        for p in self.own_id:
            self.ctx.device_free(p)
        self.maps, self.local, self.own_id = {}, None, []


class ShownFramePipeline:
    """The COMPLETE frame of the reference's loop on N GPUs -- dispatch (src/main.cpp:946, the full path tracer) then the
    display pass (:951-967, shaders/quad.frag) -- with only the DISPLAYED image crossing to rank 0: 4 bytes per pixel instead
    of the 12 of the two intermediate images.

    The display pass reads a (2R+1)^2 window, R <= 20, so a rank cannot filter interleaved 8-row tiles; here the frame is cut
    into N contiguous bands (multiples of 8 rows) and every rank traces its band plus a 20-row halo on either side, filters
    that sub-image (vrt_denoise: every tap a band pixel needs lies inside it, and where the sub-image ends early the image
    ends too, so the shader's window clipping is the same), and copies the band's displayed rows into rank 0's frame
    through an IPC mapping (vrt_device_copy), followed by a stream flag. At N = 8 a 1080p band is 135 rows + 40 of halo:
    30 % redundant tracing for a third of the traffic and a display pass that scales. Slots, streams and flags as in
    PeerFramePipeline; rank 0 is the root of every frame."""

    HALO = 20

    def __init__(self, ctx, plan_width, plan_height, rank, world, mode, n_buf=3, group=None, share=None):
        """share: rank 0's pipeline of the SAME process (tests: all ranks as objects of one process, no IPC, no collective)"""
        import numpy as np
        self.np = np
        self.ctx, self.W, self.H, self.rank, self.world, self.mode, self.n_buf, self.group = ctx, plan_width, plan_height, rank, world, mode, n_buf, group
        W, H = self.W, self.H
        tiles = (H + 7) // 8
        t0, t1 = tiles * rank // world, tiles * (rank + 1) // world
        self.b0, self.b1 = min(H, t0 * 8), min(H, t1 * 8)                   # this rank's band [b0, b1)
        self.h0, self.h1 = max(0, self.b0 - self.HALO), min(H, self.b1 + self.HALO)
        rows = self.h1 - self.h0
        self.stuck = False
        self.local, self.maps, self.sub = None, None, []
        self.streams = [torch.cuda.Stream() for _ in range(n_buf)]
        self.consumer = torch.cuda.Stream() if rank == 0 else None
        self.n_flags = (world + 1) * n_buf
        mine, err = None, None
        try:
            if self.b1 > self.b0:
                for _ in range(n_buf):   # per slot: traced rgba, (voxelID, dist), displayed rgba of the band + halo
                    self.sub.append((ctx.device_alloc(max(1, rows) * W * 4), ctx.device_alloc(max(1, rows) * W * 8), ctx.device_alloc(max(1, rows) * W * 4)))
            if rank == 0:
                self.local = {"shown": [ctx.device_alloc(W * H * 4) for _ in range(n_buf)], "flags": ctx.device_alloc(64 * self.n_flags)}
                mine = {"shown": [ctx.ipc_export(p) for p in self.local["shown"]], "flags": ctx.ipc_export(self.local["flags"])}
        except Exception as ex:  # noqa: BLE001
            err = f"rank {rank}: {type(ex).__name__}: {ex}"
        in_process = share is not None or (world > 1 and rank == 0 and group == "in-process")
        gathered = [(mine, err)]
        if world > 1 and not in_process:
            gathered = [None] * world
            dist.all_gather_object(gathered, (mine, err), group=group)
        err = next((e for _, e in gathered if e), None)
        if err is None:
            try:
                if rank == 0:
                    self.maps = self.local
                elif share is not None:
                    self.maps = share.local          # the same addresses: one process
                    self.shared = True
                else:
                    h = gathered[0][0]
                    self.maps = {"shown": [], "flags": None}
                    for x in h["shown"]:
                        self.maps["shown"].append(ctx.ipc_open(x))
                    self.maps["flags"] = ctx.ipc_open(h["flags"])
            except Exception as ex:  # noqa: BLE001
                err = f"rank {rank}: {type(ex).__name__}: {ex}"
        errs = [err]
        if world > 1 and not in_process:
            errs = [None] * world
            dist.all_gather_object(errs, err, group=group)
        err = next((e for e in errs if e), None)
        if err is not None:
            self.close()
            raise RuntimeError("shown-frame buffers: " + err)
        self.frame = 0

    shared = False

    def _arrived(self, r, k):
        return self.maps["flags"] + 64 * (r * self.n_buf + k)

    def _consumed(self, k):
        return self.maps["flags"] + 64 * (self.world * self.n_buf + k)

    def step(self):
        """enqueue one complete frame: trace band + halo, display pass, hand the band's displayed rows to rank 0"""
        ctx, W, i = self.ctx, self.W, self.frame
        k = i % self.n_buf
        stream = self.streams[k].cuda_stream
        if i >= self.n_buf:
            ctx.stream_wait_flag(self._consumed(k), i - self.n_buf + 1, stream)
        if self.b1 > self.b0:
            d_rgba, d_id, d_out = self.sub[k]
            rows = self.h1 - self.h0
            # full-frame addressing: row y of the frame lands at base + y * W * bytes, so the base is shifted up by h0 rows
            ctx.dispatch_rows(W, self.H, self.h0, self.h1, self.mode, d_rgba - self.h0 * W * 4, d_id - self.h0 * W * 8, stream)
            ctx.denoise_device(W, rows, d_rgba, d_id, d_out, stream)
            ctx.device_copy(self.maps["shown"][k] + self.b0 * W * 4, d_out + (self.b0 - self.h0) * W * 4, (self.b1 - self.b0) * W * 4, stream)
        ctx.stream_write_flag(self._arrived(self.rank, k), i + 1, stream)
        if self.rank == 0:
            cs = self.consumer.cuda_stream
            for r in range(self.world):
                ctx.stream_wait_flag(self._arrived(r, k), i + 1, cs)
            ctx.stream_write_flag(self._consumed(k), i + 1, cs)
        self.frame += 1

    def drain(self, timeout_s):
        ok = PeerFramePipeline._poll(self.streams + ([self.consumer] if self.consumer is not None else []), timeout_s)
        self.stuck = self.stuck or not ok
        return ok

    def last_shown(self):
        """rank 0: the newest displayed frame [H, W, 4] uint8"""
        if self.rank != 0 or self.frame == 0:
            return None
        return self.ctx.device_read(self.local["shown"][(self.frame - 1) % self.n_buf], (self.H, self.W, 4), self.np.uint8)

    def close(self):
        if self.stuck:
            return
        if self.maps is not None and self.maps is not self.local and not self.shared:
            for p in self.maps["shown"] + ([self.maps["flags"]] if self.maps["flags"] else []):
                self.ctx.ipc_close(p)
        if self.local:
            for p in self.local["shown"] + [self.local["flags"]]:
                self.ctx.device_free(p)
        for t in self.sub:
            for p in t:
                self.ctx.device_free(p)
        self.maps, self.local, self.sub = None, None, []
