"""PagePipeline (two batches of pages in flight: a worker thread decodes batch i-1 through a context that SHARES the weights,
cr_share_weights, while the caller prefills batch i): per page the ids are those of generate_pages, batch after batch, with and
without an EOS id, for batches of different sizes; and a context that borrows weights computes what their owner computes."""
import pytest
import torch

from callireader_amd.config import ModelDims
from callireader_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def model():
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    dims = ModelDims.reduced(vit_layers=1, llm_layers=2, rs_depth=1, vocab=8201)
    sd = synthetic.make_state_dict(dims, parts=('llm',), seed=0)
    m = InternVLChatModel.from_state_dict(sd, dims, max_tokens=768, max_pages=5)
    yield m
    m.engine.close()


def batches(seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for sizes in ((40, 170, 33), (64,), (90, 20, 300, 51, 77), (128, 128)):
        out.append([(torch.randn(S, 4096, generator=g) * 0.02).to(torch.bfloat16).cuda() for S in sizes])
    return out


@pytest.mark.parametrize('eos', [None, 'frequent'])
def test_pipeline_ids_equal_generate_pages(model, eos):
    bs = batches(3)
    new_tokens = 24
    ref = [model.generate_pages(b, max_new_tokens=new_tokens, eos_token_id=None) for b in bs]
    eos_id = None
    if eos == 'frequent':                                   # an id that some pages do emit: their rows leave the batch early
        flat = [t for o in ref for ids in o for t in ids[3:]]
        eos_id = max(set(flat), key=flat.count)
        ref = [model.generate_pages(b, max_new_tokens=new_tokens, eos_token_id=eos_id, check_every=4) for b in bs]
        assert any(len(ids) < new_tokens for o in ref for ids in o)
    pipe = model.page_pipeline(max_new_tokens=new_tokens, eos_token_id=eos_id, check_every=4)
    outs = []
    for b in bs:
        prev = pipe.start(b)
        if prev is not None:
            outs.append(prev)
    outs.append(pipe.finish())
    assert pipe.finish() is None                             # nothing left in flight
    pipe.close()
    assert outs == ref


def test_borrowed_weights_compute_what_the_owner_computes(model):
    from callireader_amd.engine import Engine
    eng = model.engine
    other = Engine(eng.dims, device=eng.device.index, max_pos=eng.max_pos)
    other.share_weights_from(eng)
    g = torch.Generator().manual_seed(9)
    emb = (torch.randn(200, 4096, generator=g) * 0.02).to(torch.bfloat16).cuda()
    kv = eng.kv_alloc(2, 512)
    a = eng.prefill(kv, 0, emb, want_logits=True)
    b = other.prefill(kv, 1, emb, want_logits=True)         # same cache object, filled through the other context
    s1 = eng.decode(kv, [1], want_logits=True)              # ... and read back through the owner
    s0 = other.decode(kv, [0], want_logits=True)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(s0, s1)
    kv.free()
    from callireader_amd._binding import CalliReaderError
    with pytest.raises(CalliReaderError):                   # a borrower cannot load, finalize or build fp8 copies
        other.load_weight('language_model.model.norm.weight', torch.ones(4096, dtype=torch.bfloat16))
    with pytest.raises(CalliReaderError):
        other.enable_fp8_decode(True)
    other.close()                                           # does not free the owner's weights
    c = eng.prefill(eng.kv_alloc(1, 512), 0, emb, want_logits=True)
    torch.cuda.synchronize()
    assert torch.equal(a, c)


def test_pipeline_caches_grow_only_when_idle(model):
    """Batches that outgrow the pipeline's KV caches (1, 3, 5, 2 pages with caches that start at one page): a cache is regrown
    only when its own turn comes -- never the one the worker thread is decoding from -- and every batch keeps its ids."""
    bs = batches(5)
    bs = [bs[1], bs[0], bs[2], bs[3]]
    ref = [model.generate_pages(b, max_new_tokens=20, eos_token_id=None) for b in bs]
    old = model.max_pages
    model.max_pages = 1
    try:
        pipe = model.page_pipeline(max_new_tokens=20, eos_token_id=None, check_every=4)
        outs = []
        for b in bs:
            prev = pipe.start(b)
            if prev is not None:
                outs.append(prev)
        outs.append(pipe.finish())
        assert [kv.n_seqs for kv in pipe.kvs] == [5, 3]
        pipe.close()
    finally:
        model.max_pages = old
    assert outs == ref


def test_borrower_notices_that_the_owner_changed(model):
    """cr_share_weights copies a snapshot of the owner's tensor map: once the owner builds fp8 copies, switches an fp8 option or
    reloads, the borrower's stage entry points refuse to run until the weights are shared again."""
    from callireader_amd.engine import Engine
    from callireader_amd._binding import CalliReaderError
    eng = model.engine
    other = Engine(eng.dims, device=eng.device.index, max_pos=eng.max_pos)
    other.share_weights_from(eng)
    g = torch.Generator().manual_seed(11)
    emb = (torch.randn(100, 4096, generator=g) * 0.02).to(torch.bfloat16).cuda()
    kv = eng.kv_alloc(2, 256)
    a = other.prefill(kv, 0, emb, want_logits=True).clone()
    eng.enable_fp8_decode(True)
    try:
        with pytest.raises(CalliReaderError, match='share again'):
            other.prefill(kv, 1, emb)
        with pytest.raises(CalliReaderError, match='share again'):
            other.decode(kv, [0])
    finally:
        eng.enable_fp8_decode(False)
    with pytest.raises(CalliReaderError, match='share again'):      # switching it off changes the owner's generation too
        other.prefill(kv, 1, emb)
    other.share_weights_from(eng)
    b = other.prefill(kv, 1, emb, want_logits=True)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    kv.free()
    other.close()


def test_borrower_outlives_its_owner_safely():
    """The borrower watches a generation cell that outlives the owning context (round-3 advisor finding: it used to dereference the owner):
    once the owner is destroyed every stage entry point of the borrower -- and of a borrower of the borrower -- refuses to run."""
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    from callireader_amd.engine import Engine
    from callireader_amd._binding import CalliReaderError
    dims = ModelDims.reduced(vit_layers=1, llm_layers=1, rs_depth=1, vocab=8201)
    sd = synthetic.make_state_dict(dims, parts=('llm',), seed=1)
    owner = InternVLChatModel.from_state_dict(sd, dims, max_tokens=256, max_pages=1)
    eng = owner.engine
    b1 = Engine(eng.dims, device=eng.device.index, max_pos=eng.max_pos)
    b1.share_weights_from(eng)
    b2 = Engine(eng.dims, device=eng.device.index, max_pos=eng.max_pos)
    b2.share_weights_from(b1)                               # a borrower of a borrower watches the real owner
    g = torch.Generator().manual_seed(12)
    emb = (torch.randn(40, 4096, generator=g) * 0.02).to(torch.bfloat16).cuda()
    kv = b2.kv_alloc(1, 128)
    a = eng.prefill(kv, 0, emb, want_logits=True).clone()
    kv.reset()
    b = b2.prefill(kv, 0, emb, want_logits=True).clone()
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    eng.enable_fp8_decode(True)                             # the owner changes: both levels of borrowing notice
    for e in (b1, b2):
        with pytest.raises(CalliReaderError, match='share again'):
            e.prefill(kv, 0, emb)
    b3 = Engine(eng.dims, device=eng.device.index, max_pos=eng.max_pos)
    with pytest.raises(CalliReaderError, match='share again'):      # a stale borrower hands on nothing
        b3.share_weights_from(b1)
    eng.enable_fp8_decode(False)
    b1.share_weights_from(eng); b2.share_weights_from(b1)
    kv.reset()
    b2.prefill(kv, 0, emb)
    torch.cuda.synchronize()
    eng.close()                                             # cr_destroy frees the tensors
    for e in (b1, b2):
        with pytest.raises(CalliReaderError, match='destroyed'):
            e.prefill(kv, 0, emb)
    kv.free()
    for e in (b1, b2, b3):
        e.close()
