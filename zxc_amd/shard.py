"""Per-GPU block ranges for multi-GPU decode: blocks are independent, so
rank g of G owns the contiguous index range [g*N//G, (g+1)*N//G) — contiguous keeps both its
compressed input and its output a single span. No collective touches the data path."""


def block_range(rank: int, world: int, n_blocks: int):
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * n_blocks) // world, ((rank + 1) * n_blocks) // world

