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


def test_young_moving_or_nearly_full_kinds_are_checked_in_the_forward_call():
    """What the key of a kind cannot see is the camera's distance.  While a kind is young, while its counts move by more than 1.5x
    between frames, or while the last frame used more than a third of the bound, the forward call looks at the status words itself
    (and renders again what did not fit); only steady kinds are left to the late check in backward()."""
    book = fv.CapacityBook()
    k = ("cuda:0", 512, 512, 100_000, 0, 0)
    assert book.check_early(k)                                                               # never seen
    for i in range(fv.CapacityBook.SETTLE - 1):
        book.learn(k, 500_000)
        assert book.check_early(k)                                                           # young
    book.learn(k, 510_000)
    assert not book.check_early(k)                                                           # four steady frames, 8x headroom
    book.learn(k, 900_000)                                                                   # the camera walks in: 1.76x in one frame
    assert book.check_early(k)
    for _ in range(fv.CapacityBook.SETTLE):
        book.learn(k, 900_000)
    assert not book.check_early(k)                                                           # steady again
    for n in (1_200_000, 1_500_000, 1_900_000, 2_400_000):                                   # a slow walk: < 1.5x per frame ...
        book.learn(k, n)
    assert 3 * 2_400_000 > book.get(k) or not book.check_early(k)
    assert book.get(k) >= fv.CapacityBook.MARGIN * 900_000
    book.bound[k] = 3 * 2_400_000 - 1                                                        # ... but the headroom is short: checked
    assert book.check_early(k)
    book.reset()
    assert book.get(k) is None and book.check_early(k)


def test_frames_rendered_with_gradients_but_never_differentiated_still_teach_the_book():
    book_before, hist_before = dict(fv.capacity_book.bound), dict(fv.capacity_book.history)
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
        fv.capacity_book.reset()
        fv.capacity_book.bound.update(book_before)
        fv.capacity_book.history.update(hist_before)


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
        fv.capacity_book.history.pop(("cpu", 64, 48, 11, 0, 0), None)
        fv._pinned_free[:] = free_before
        fv._pinned_limbo[:] = limbo_before
