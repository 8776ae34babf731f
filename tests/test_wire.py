"""The serialised form of a trajectory ("DPTJ" v1, include/dftpav_hip.h): pure host code of the
C-ABI library, so these run without a GPU."""
import struct

import numpy as np
import pytest

from dftpav_amd import pods


@pytest.fixture(scope="module")
def capi():
    from dftpav_amd import capi as c
    c.lib()
    return c


def _traj(rng, piece_nums):
    n = int(np.sum(piece_nums))
    return rng.normal(size=(n, 6, 2)), rng.uniform(0.3, 2.0, size=len(piece_nums))


def test_round_trip_is_bit_exact(capi):
    rng = np.random.default_rng(5)
    lay = pods.LayoutSpec([3, 2, 4], [1, -1, 1])
    co, dt = _traj(rng, lay.piece_nums)
    blob = capi.wire_pack(lay, co, dt, drone_id=7, traj_id=42, start_time=12.5)
    assert len(blob) == capi.wire_size(lay.piece_nums) == 32 + 3 * 24 + 9 * 104
    assert blob[:4] == b"DPTJ" and struct.unpack_from("<HBB", blob, 4) == (1, 5, 2)
    u = capi.wire_unpack(blob)
    assert (u["drone_id"], u["traj_id"], u["start_time"]) == (7, 42, 12.5)
    assert list(u["singuls"]) == [1, -1, 1] and list(u["piece_nums"]) == [3, 2, 4]
    # highest power first, x then y: column 0 of CoefficientMat multiplies t^5 (poly_traj_utils.hpp:77-87)
    assert np.array_equal(u["coeffs"].reshape(9, 6, 2), co[:, ::-1, :])
    assert np.array_equal(u["durations"], np.repeat(dt, lay.piece_nums))
    # segment times as addSingulTraj chains them: duration = running sum of the piece durations, start = previous end
    world = 12.5
    for i, N in enumerate(lay.piece_nums):
        d = 0.0
        for _ in range(N):
            d += dt[i]
        assert u["seg_start"][i] == world and u["seg_duration"][i] == d
        world = world + d


def test_bad_blobs_are_refused(capi):
    rng = np.random.default_rng(6)
    lay = pods.LayoutSpec([2], [1])
    co, dt = _traj(rng, lay.piece_nums)
    blob = capi.wire_pack(lay, co, dt)
    for bad in (blob[:-1], b"XXXX" + blob[4:], blob[:4] + b"\x02\x00" + blob[6:], blob[:31], b""):
        with pytest.raises(capi.DftpavError):
            capi.wire_unpack(bad)
    # a segment claiming more pieces than the blob holds
    forged = bytearray(blob)
    struct.pack_into("<i", forged, 32 + 4, 3)
    with pytest.raises(capi.DftpavError):
        capi.wire_unpack(bytes(forged))
    assert capi.wire_size([0]) == 0 and capi.wire_size([]) == 0
