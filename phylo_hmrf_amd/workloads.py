"""Block lists of the BASELINE.json configurations (synthetic shapes; no reference file is read at run time).

hg38 autosome lengths are public genome facts (UCSC hg38.chrom.sizes); the chr3 / chr6 centromere split points are
the ones the reference hard-codes (utility.py:385).  A chromosome is one diagonal block (upper triangle of an
N x N contact map, N = ceil(size / resolution)); chr3 and chr6 give two diagonal blocks and one off-diagonal block.
"""
import math

HG38_AUTOSOMES = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
                  133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
                  58617616, 64444167, 46709983, 50818468]
SPLIT = {3: (90279522, 93797661), 6: (57542947, 61520508)}     # utility.py:385


def genome_blocks(resolution):
    """-> list of (H, W, diagonal) for the 22 autosomes at the given bin size."""
    out = []
    for c, size in enumerate(HG38_AUTOSOMES, start=1):
        if c in SPLIT:
            a, b = SPLIT[c]
            n1 = int(math.ceil(a / float(resolution)))
            n2 = int(math.ceil((size - b) / float(resolution)))
            out += [(n1, n1, True), (n2, n2, True), (n1, n2, False)]
        else:
            n = int(math.ceil(size / float(resolution)))
            out.append((n, n, True))
    return out


def block_nodes(H, W, diagonal):
    return H * (H + 1) // 2 if diagonal else H * W


WORKLOADS = {
    # name: (blocks, S, K, num_neighbor, description)
    "cfg2": ([(2000, 2000, True)], 4, 10, 8, "synthetic 1 chrom, 2000x2000 bin-pairs (upper triangle, 2,001,000 nodes), 4 species, K=10, num_neighbor=8"),
    "cfg3": (None, 4, 20, 8, "synthetic whole-genome hg38 50 kb, 4 species, K=20 (num_neighbor 10 -> 8: the reference raises KeyError for 10, utility.py:1909-1916)"),
    "cfg4": (None, 8, 30, 8, "synthetic whole-genome 50 kb, 8 species (balanced 8-leaf tree), K=30"),
    # BASELINE configs[4]: 10 kb bins.  2.23e9 nodes in 26 blocks, the chr1 block alone 310 M nodes (121 GB at 390 B/node):
    # the whole workload needs >= 4 GPUs (blocks sharded by dist.lpt_assign); "cfg5-chr1" is its largest block alone,
    # the biggest single MRF the path has to hold on one GPU.
    "cfg5": ("10kb", 4, 20, 8, "synthetic whole-genome hg38 10 kb, 4 species, K=20, syntenic blocks sharded over the GPUs"),
    "cfg5-chr1": ([(24896, 24896, True)], 4, 20, 8, "the chr1 block of the 10 kb workload alone (309,917,856 nodes), 4 species, K=20"),
    # what the fullest rank of an 8-GPU strong-scaling run of cfg3 holds: the chr1 block alone
    "cfg3-chr1": ([(4980, 4980, True)], 4, 20, 8, "the chr1 block of the 50 kb workload alone (12,402,690 nodes), 4 species, K=20"),
    "small": ([(300, 300, True), (200, 260, False)], 4, 20, 8, "small smoke workload"),
}

BYTES_PER_NODE = 390          # resident HBM per node at K=20, S=4


def workload(name):
    blocks, S, K, nn, desc = WORKLOADS[name]
    if blocks is None:
        blocks = genome_blocks(50000)
    elif blocks == "10kb":
        blocks = genome_blocks(10000)
    return blocks, S, K, nn, desc


def shard(blocks, world, scaling="strong"):
    """Which blocks each rank owns.  strong: the workload's blocks are dealt to the ranks longest-processing-time-first
    (dist.lpt_assign), total work fixed -- north_star's "nodes shard naturally by syntenic block across the GPUs";
    weak: the workload is replicated `world` times (a `world`-genome cohort) and dealt the same way, so the work per
    GPU stays that of one copy.  -> (all_blocks, owner[len(all_blocks)])"""
    from .dist import lpt_assign
    all_blocks = list(blocks) * (world if scaling == "weak" else 1)
    owner = lpt_assign([block_nodes(*b) for b in all_blocks], world)
    return all_blocks, owner
