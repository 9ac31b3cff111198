"""Workload of bench.py: the page shape, the synthetic prompt ids, and which pages / character tiles a rank handles in a step (pure host arithmetic)."""
import torch

NEW_TOKENS = 128               # greedy tokens per page (bench.py --new-tokens sets it: plan.NEW_TOKENS)
PAGE_TILES, CHAR_TILES, TEXT_TOKENS = 11, 96, 60


def build_ids(n_page_tiles, n_char_tiles, text_tokens, img_id, ref_id, seed):
    g = torch.Generator().manual_seed(seed)
    head = torch.randint(100, 60000, (text_tokens // 2,), generator=g)
    tail = torch.randint(100, 60000, (text_tokens - text_tokens // 2,), generator=g)
    return torch.cat([head, torch.full((n_page_tiles * 256,), img_id), torch.full((n_char_tiles * 3,), ref_id), tail])


def plan_workload(scaling, pages, total_pages, world, rank, plan='even', cost=None, owners=None):
    """Which pages and which character tiles one rank handles in a step.  weak: `pages` per GPU whatever N (n_pages = pages * world);
    strong: `total_pages` per step over all ranks (BASELINE config 4 as written: 64 pages over 8 GPUs).  plan 'even': pages are owned
    round-robin, the flat list of character tiles is split contiguously and evenly; plan 'balanced' (strong only): fewer ranks own pages
    and the others take more character tiles (callireader_amd/parallel.py: plan_balanced).  Pure host arithmetic."""
    from callireader_amd.parallel import shard_range, owned_pages, plan_balanced
    if scaling not in ('weak', 'strong') or plan not in ('even', 'balanced'):
        raise ValueError((scaling, plan))
    n_pages = total_pages if scaling == 'strong' else pages * world
    if plan == 'balanced':
        if scaling != 'strong':
            raise ValueError('the balanced plan is a strong-scaling plan')
        from callireader_amd.parallel import MI355X_COST
        pb = plan_balanced(n_pages, world, PAGE_TILES, CHAR_TILES, PAGE_TILES * 256 + CHAR_TILES * 3 + TEXT_TOKENS, NEW_TOKENS, cost=cost or MI355X_COST, owners=owners)
        lo, hi = pb['char_bounds'][rank]
        return {'scaling': scaling, 'plan': 'balanced', 'n_pages': n_pages, 'mine': pb['pages'][rank], 'ct_lo': lo, 'ct_hi': hi, 'ct_counts': pb['char_counts'],
                'pages_per_gpu': len(pb['pages'][rank]), 'balanced': pb}
    if n_pages < world:
        raise SystemExit(f'{n_pages} pages per step < {world} ranks: every rank needs a page')
    mine = owned_pages(n_pages, world, rank)
    ct_lo, ct_hi = shard_range(n_pages * CHAR_TILES, world, rank)
    return {'scaling': scaling, 'plan': 'even', 'n_pages': n_pages, 'mine': mine, 'ct_lo': ct_lo, 'ct_hi': ct_hi, 'pages_per_gpu': len(mine)}


def plan_strong_share(total_pages, share_world):
    """The workload `strong_share` runs on ONE GPU: as much as rank 0 of plan_workload('strong', total_pages over share_world ranks) has -- the same
    number of owned pages and of character tiles -- but laid on pages 0 .. n-1 of the one-GPU step and their own character tiles (a rank of the real
    run owns pages r, r + world, ... and an arbitrary eighth of the character tiles: the same work), so that the ids can be compared with the
    full step's.  Needs an even split (total_pages % share_world == 0)."""
    sw = plan_workload('strong', total_pages, total_pages, share_world, 0)
    n_own, n_ct = sw['pages_per_gpu'], sw['ct_hi'] - sw['ct_lo']
    if n_ct != n_own * CHAR_TILES:
        raise ValueError(f'{total_pages} pages over {share_world} ranks: uneven split')
    return {'scaling': 'strong', 'n_pages': n_own, 'mine': list(range(n_own)), 'ct_lo': 0, 'ct_hi': n_ct, 'pages_per_gpu': n_own}
