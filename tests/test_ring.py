"""RingBuffer2D bookkeeping.  These are the ONLY results the reference's own tests pin
(tests/RingBufferTest.h:150-283): constructor capacities, push/size, mark() == (i+1) % capacity,
reservation dirtiness after a full wrap.  Restated here against the oracle's ring
(ro_oracle_ring2d_*) and, when built, the product's host-side ring (host/ring_capi)."""
import ctypes as C

import pytest


class OracleRing:
    def __init__(self, oracle, width, chunk, capacity=None):
        self.L = oracle.lib()
        self.h = self.L.ro_oracle_ring2d_create(4, width, chunk, -1 if capacity is None else capacity)

    def __getattr__(self, name):
        fn = getattr(self.L, "ro_oracle_ring2d_" + name)
        return lambda *a: fn(self.h, *a)


@pytest.fixture(params=["oracle", "host"])
def ring_factory(request, oracle):
    if request.param == "oracle":
        return lambda w, c, cap=None: OracleRing(oracle, w, c, cap)
    from hostlib import HostRing, host_library
    if host_library() is None:
        pytest.skip("host library not built")
    return lambda w, c, cap=None: HostRing(w, c, cap)


SZ = 4   # sizeof(int), as in the reference's RingBuffer2D<int> tests


@pytest.mark.parametrize("chunk", [SZ * 16 * 8, SZ * 16 * 8 - SZ])
def test_constructor_without_capacity(ring_factory, chunk):      # RingBufferTest.h:170-178, :191-195
    r = ring_factory(16, chunk)
    assert r.capacity() == 0 and r.get_size() == 0 and not r.is_full()


@pytest.mark.parametrize("chunk", [SZ * 16 * 8, SZ * 16 * 8 - SZ])
def test_constructor_with_capacity(ring_factory, chunk):         # :180-189, :197-201
    r = ring_factory(16, chunk, 16 * 8 * 8)
    assert r.capacity() >= 16 * 8 * 8 and r.get_size() == 0 and not r.is_full()
    assert r.chunk_rows() == 8                                   # ceil(chunk / rowBytes), RingBuffer.h:436-438


@pytest.mark.parametrize("chunk", [SZ * 16 * 8, SZ * 16 * 8 - 1])
def test_push_size(ring_factory, chunk):                         # :203-226
    cap = 16 * 8 * 8
    r = ring_factory(16, chunk, cap)
    for i in range(cap * 3):
        r.push()
        assert r.get_size() == (r.capacity() if i + 1 > r.capacity() else i + 1)


@pytest.mark.parametrize("chunk", [SZ * 16 * 8, SZ * 16 * 8 - 1])
def test_mark_is_next_row(ring_factory, chunk):                  # :228-246
    cap = 16 * 8 * 8
    r = ring_factory(16, chunk, cap)
    for i in range(cap * 3):
        r.push()
        assert r.mark() == (i + 1) % r.capacity()


@pytest.mark.parametrize("chunk,cap", [(SZ * 16 * 8, 16 * 8 * 8), (SZ * 16 * 8 - 1, 16 * 8 * 8),
                                       (SZ * 16 * 8, 16 * 8 * 31), (SZ * 16 * 8 - 1, 16 * 8 * 31)])
def test_reservations_get_dirty_after_a_wrap(ring_factory, chunk, cap):   # :248-279
    r = ring_factory(16, chunk, cap)
    mark = r.mark()
    for _ in range(10):
        r.push()
    assert r.size_from(mark) == 10
    h = r.reserve(mark, 5)
    assert r.is_dirty(h) == 0
    for _ in range(cap):
        r.push()
    assert r.is_dirty(h) == 1


def test_ring_quirks_used_by_recorders(ring_factory):
    """at()/size() semantics recorders rely on (src/RingBuffer.h:360-369, :543-560)."""
    r = ring_factory(32768, 1024 * 1024, 2816)                   # C3 default ring: 352*8 rows, 8 rows/chunk
    assert r.capacity() == 2816 and r.chunk_rows() == 8
    assert r.size_from(r.mark()) == r.capacity()                 # start == head -> capacity (Appendix B-7)
    for _ in range(5):
        r.push()
    assert r.normalize(-1) == 2815 and r.normalize(r.mark() - 1) == 4
    assert r.size_from(0) == 5 and r.size_between(2810, 4) == 10
    h = r.reserve(0, 3)
    assert r.free_reservation(h) == 1 and r.free_reservation(99) == 0
    assert r.reserve(1, 2) == h                                  # handles are recycled (:589-591)


def test_a_run_of_pushes_is_n_pushes():
    """RingBuffer2D::pushRun / pushWritten / markAhead of the host mirror (what makes Backend::process cost per call, not
    per sample, and lets the GPU write rows into the ring's slots): head, size, the slots handed out and every
    reservation's dirty flag are those of n single push() calls -- random rings, random reservations (with the
    reference's end = 0 quirk in them), runs that wrap."""
    import random
    from hostlib import HostRing, host_library
    if host_library() is None:
        pytest.skip("tests/harness/libro_host_harness.so is not built")
    rng = random.Random(7)
    for trial in range(300):
        width = rng.choice([1, 2, 5])
        cap_req = rng.randint(3, 40)
        a, b = HostRing(width, 4 * width * rng.choice([1, 3, 4]), cap_req), None
        b = HostRing(width, 1, 1)
        b = HostRing(width, 4 * width, 1)
        # two rings of the same geometry
        chunk = 4 * width * rng.choice([1, 3, 4])
        a, b = HostRing(width, chunk, cap_req), HostRing(width, chunk, cap_req)
        cap = a.capacity()
        assert b.capacity() == cap
        pre = rng.randint(0, 2 * cap)
        for i in range(pre):
            a.push(); b.push()
        hs = []
        for _ in range(rng.randint(0, 4)):
            s_, e_ = rng.randint(-cap, 2 * cap), rng.randint(-cap, 2 * cap)
            hs.append((a.reserve(s_, e_), b.reserve(s_, e_)))
        for step in range(rng.randint(1, 6)):
            n = rng.randint(0, cap + 3)
            kind = rng.choice(["run", "written", "ahead"])
            if kind == "ahead":
                # markAhead(n) = the dirty flags n pushes would set, nothing else
                a.mark_ahead(n)
                c = HostRing(width, chunk, cap_req)          # a scratch copy of b's state to push into
                for i in range(b.mark() if b.get_size() < cap else cap + b.mark()):
                    c.push()
                assert c.mark() == b.mark()
                want_dirty = []
                for (ha, hb) in hs:
                    want_dirty.append(b.is_dirty(hb))
                # replay on b itself, then compare flags only (b's head moves; a's must not)
                m_a = a.mark()
                for i in range(min(n, cap)):
                    b.push()
                for (ha, hb) in hs:
                    assert a.is_dirty(ha) == b.is_dirty(hb), (trial, step, kind, n)
                assert a.mark() == m_a
                a.push_written(min(n, cap))                  # bring a level with b again
            else:
                m = a.push_run(n, 1000.0 * step) if kind == "run" else (a.mark(), a.push_written(n))[0]
                for i in range(n):
                    assert b.mark() == (m + i) % cap
                    b.push()
                    if kind == "run" and i >= n - cap:
                        assert a.at0((m + i) % cap) == 1000.0 * step + i
            assert a.mark() == b.mark() and a.get_size() == b.get_size() and a.is_full() == b.is_full()
            for (ha, hb) in hs:
                assert a.is_dirty(ha) == b.is_dirty(hb), (trial, step, kind, n)
