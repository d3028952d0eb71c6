"""Host logic of the one-call view path (soar_amd/renderer/fused_view.py) that needs no GPU: the capacity book that sizes binning
buffers from earlier frames, and what is learnt from the status words of frames nobody differentiated."""
import math
import types
import warnings

import torch

from soar_amd.renderer import fused_view as fv


def _rs(W=1920, H=1080, fovx=0.8, fovy=0.5):
    return types.SimpleNamespace(image_width=W, image_height=H, tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2))


def test_a_kind_is_device_size_model_and_field_of_view_to_a_quarter_octave():
    k = fv.CapacityBook.key("cuda:0", _rs(), 100_000)
    assert k[:4] == ("cuda:0", 1920, 1080, 100_000)
    assert fv.CapacityBook.key("cuda:0", _rs(fovx=0.8 * 1.02), 100_000) == k                 # 2 % more field of view: same kind
    assert fv.CapacityBook.key("cuda:0", _rs(fovx=0.8 * 1.5), 100_000) != k                  # a zoom is another kind
    assert fv.CapacityBook.key("cuda:0", _rs(), 120_000) != k and fv.CapacityBook.key("cuda:1", _rs(), 100_000) != k
    assert fv.CapacityBook.key("cuda:0", _rs(W=1024, H=1024), 100_000) != k


def test_bounds_grow_before_they_are_used_up_and_never_shrink():
    book = fv.CapacityBook()
    k = ("cuda:0", 160, 120, 3000, -2, -5)
    assert book.get(k) is None
    assert book.learn(k, 5000) == fv.CapacityBook.FLOOR                                      # small scenes get the floor
    assert book.learn(k, 700_000) == fv.CapacityBook.MARGIN * 700_000
    bound = book.get(k)
    assert book.learn(k, bound // 2) == bound and book.learn(k, 10) == bound                  # half used / hardly used: unchanged
    assert book.learn(k, bound // 2 + 1) == fv.CapacityBook.MARGIN * (bound // 2 + 1)         # more than half used: grown
    assert book.learn(k, 3) == book.get(k)


def test_frames_rendered_with_gradients_but_never_differentiated_still_teach_the_book():
    book_before = dict(fv.capacity_book.bound)
    try:
        k = ("cpu", 64, 48, 10, 0, 0)
        fv.capacity_book.bound[k] = 1 << 20
        rs = types.SimpleNamespace(image_width=64, image_height=48)
        words = torch.tensor([700_000, 0, 0, 0], dtype=torch.int32)
        p = fv._PendingStatus([(rs, None, 1 << 20, k, False)], [words], "cpu", None)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            del p                                                     # fitted: learnt, silent
        assert fv.capacity_book.bound[k] == fv.CapacityBook.MARGIN * 700_000
        words = torch.tensor([0, 9_000_000, 0, 0], dtype=torch.int32)      # did not fit: 9 M needed
        p = fv._PendingStatus([(rs, None, fv.capacity_book.bound[k], k, False)], [words], "cpu", None)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            del p
        assert any("rendered as background" in str(x.message) for x in w)
        assert fv.capacity_book.bound[k] == fv.CapacityBook.MARGIN * 9_000_000
        words = torch.tensor([-1, -1, 0, 0], dtype=torch.int32)            # copy not landed, nobody waits: nothing learnt, nothing said
        p = fv._PendingStatus([(rs, None, 5, k, False)], [words], "cpu", None)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            del p
    finally:
        fv.capacity_book.bound.clear()
        fv.capacity_book.bound.update(book_before)


def test_status_words_whose_copy_may_still_land_are_never_handed_out_again():
    """The device-to-host copy into a view's status words is issued by the C library: torch's host allocator knows nothing of it.
    Words nobody waited for (outputs rendered with gradients and dropped) must stay referenced -- and out of the free list -- until
    their sentinels are gone; releasing a block twice (error paths come by twice) must not put it on a list twice."""
    free_before, limbo_before = list(fv._pinned_free), list(fv._pinned_limbo)
    try:
        fv._pinned_free.clear(); fv._pinned_limbo.clear()
        mk = lambda v: torch.tensor(v, dtype=torch.int32)
        landed, flying = mk([5, 0, 0, 0]), mk([-1, -1, 0, 0])
        rs = types.SimpleNamespace(image_width=64, image_height=48)
        k = ("cpu", 64, 48, 11, 0, 0)
        p = fv._PendingStatus([(rs, None, 1 << 20, k, False), (rs, None, 1 << 20, k, False)], [landed, flying], "cpu", None)
        del p                                                         # end of life without a backward pass
        assert any(w is landed for w in fv._pinned_free) and not any(w is flying for w in fv._pinned_free)
        assert any(w is flying for w in fv._pinned_limbo)             # still referenced: its block cannot be recycled
        fv._release_words([landed, flying])                           # a second release changes nothing
        assert len(fv._pinned_free) == 1 and len(fv._pinned_limbo) == 1
        got = fv._status_words()                                      # the copy has not landed: the flying words are not handed out
        assert got is landed and len(fv._pinned_limbo) == 1
        flying[:2] = 3                                                # ... now it has
        fv._release_words([got])
        a, b = fv._status_words(), fv._status_words()
        assert {id(a), id(b)} == {id(landed), id(flying)} and not fv._pinned_limbo
    finally:
        fv.capacity_book.bound.pop(("cpu", 64, 48, 11, 0, 0), None)
        fv._pinned_free[:] = free_before
        fv._pinned_limbo[:] = limbo_before
