"""Row sharding of one frame over the GPUs of a node and the gather back to rank 0.

Pixels are independent, so the frame is cut into tiles of `tile_rows` rows dealt round-robin to the
ranks (sky rows are several times cheaper than model rows; interleaving balances them). Each rank
traces its tiles into ONE compact device buffer laid out as

    [ rgba8 block : rows_max * W int32 ][ id/dist block : rows_max * W * 2 int32 ]

so the only exchange step of the path -- the gather of finished rows to rank 0 -- is a single
collective per frame (RCCL over xGMI with backend "nccl", gloo in the CPU tests). Rank 0 then
un-interleaves the rows into frame order with one index_copy per image.
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

    def frame_index(self, device):
        """for rank 0: per source rank, the frame rows its compact rows land on"""
        return [torch.tensor(r, dtype=torch.long, device=device) for r in self.rows_of]


def gather_frame(plan, local, gathered, frame_rgba, frame_id, row_index, group=None):
    """Gathers every rank's buffer to rank 0 and scatters the rows into frame order there.

    local: this rank's buffer; gathered: (rank 0) list of `world` buffers or None;
    frame_rgba [H, W] int32, frame_id [H, W, 2] int32 (rank 0)."""
    if plan.world > 1:
        dist.gather(local, gathered if plan.rank == 0 else None, dst=0, group=group)
    else:
        gathered = [local]
    if plan.rank != 0:
        return
    W, rm = plan.width, plan.rows_max
    for r, buf in enumerate(gathered):
        n = len(plan.rows_of[r])
        if n == 0:
            continue
        frame_rgba.index_copy_(0, row_index[r], buf[: rm * W].view(rm, W)[:n])
        frame_id.index_copy_(0, row_index[r], buf[rm * W:].view(rm, W, 2)[:n])
