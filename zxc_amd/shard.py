"""Per-GPU block ranges for multi-GPU decode: blocks are independent, so
rank g of G owns the contiguous index range [g*N//G, (g+1)*N//G) — contiguous keeps both its
compressed input and its output a single span. No collective touches the data path."""


def block_range(rank: int, world: int, n_blocks: int):
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * n_blocks) // world, ((rank + 1) * n_blocks) // world


def byte_balanced_ranges(world: int, costs):
    """Contiguous split by cumulative cost (comp_size + decomp_size per block) instead of count."""
    total = float(sum(costs))
    bounds = [0]
    acc = 0.0
    g = 1
    for i, c in enumerate(costs):
        acc += c
        while g < world and acc >= total * g / world:
            bounds.append(i + 1)
            g += 1
    while len(bounds) < world:
        bounds.append(len(costs))
    bounds.append(len(costs))
    return [(bounds[g], bounds[g + 1]) for g in range(world)]
