"""Host-side configuration contract: INI -> VFO tree (mainwindow.cpp:27-233) and the BASELINE
workload generators."""
import os

import numpy as np
import pytest

from helpers import REFERENCE_ROOT
from sdrreceiver_amd import synth, topology as tp

INI_25E_LIKE = """
sample_rate=1536000
center_frequency=1545600000
zmq_address=tcp://*:6003
correct_dc_bias=1
mix_offset=0
#remote_rtl=127.0.0.1:1234

[main_vfos]
size=2
1\\frequency=1545116000
1\\out_rate=384000
2\\frequency=1546096000
2\\out_rate=192000

[vfos]
size=3
1\\frequency=1545005146
1\\gain=5
1\\filter_bandwidth=4000
1\\data_rate=600
1\\topic=VFO01
2\\frequency=1545124261
2\\gain=5
2\\data_rate=1200
2\\fiter_bandwidth=9999
2\\topic=VFO07
3\\frequency=1546137300
3\\gain=3
3\\data_rate=10500
3\\filter_bandwidth=10000
3\\topic=VFO19
"""


def test_ini_rules_25e_like():
    t = tp.topology_from_ini(INI_25E_LIKE)
    assert (t.fs, t.frame, t.bufsplit, t.correct_dc) == (1536000, 384000, 4, True)
    m0, m1, a, b, c = t.vfos
    assert (m0.decimate_count, m0.mixer_freq, m0.demod_usb, m0.samples_per_buffer) == (2, 484000.0, False, 384000)
    assert (m1.decimate_count, m1.mixer_freq) == (3, -496000.0)
    assert (a.parent, a.fs, a.decimate_count, a.mixer_freq, a.filter_bw, a.samples_per_buffer) == (0, 384000, 5, 110854.0, 4000, 96000)
    assert a.gain == float(np.float32(5) / np.float32(100)) and a.output_rate == 12000 and a.n_out == 3000
    assert (b.parent, b.decimate_count, b.filter_bw, b.output_rate) == (0, 4, 0, 24000)  # misspelt key ignored
    assert (c.parent, c.fs, c.decimate_count, c.mixer_freq, c.filter_bw, c.output_rate) == (1, 192000, 2, -41300.0, 10000, 48000)


def test_ini_rules_with_a_mix_offset():
    """mainwindow.cpp:65,151: `mix_offset` is added to every SUB VFO's frequency before anything is derived from it -- its mixer
    (centre - main - frequency) and the choice of its main VFO -- and to no main VFO's (mainwindow.cpp:106)."""
    base = tp.topology_from_ini(INI_25E_LIKE)
    for off in (1500, -2500):
        t = tp.topology_from_ini(INI_25E_LIKE.replace("mix_offset=0", f"mix_offset={off}"))
        assert [(v.mixer_freq, v.decimate_count) for v in t.vfos[:2]] == [(v.mixer_freq, v.decimate_count) for v in base.vfos[:2]]
        for v, b in zip(t.vfos[2:], base.vfos[2:]):
            assert v.mixer_freq == b.mixer_freq - off
            assert (v.parent, v.fs, v.decimate_count, v.filter_bw, v.gain, v.samples_per_buffer) == \
                (b.parent, b.fs, b.decimate_count, b.filter_bw, b.gain, b.samples_per_buffer)
    # an offset that carries a sub VFO out of its main's band and into the other main's: |main - (f + offset)| < out_rate picks
    # the FIRST main that covers it (mainwindow.cpp:179-191).  VFO19 sits 41.3 kHz above main 2 (1 546 096 000, 192 k wide) and
    # 1 021.3 kHz above main 1 (384 k wide): -700 000 Hz puts it 321.3 kHz above main 1 -> inside main 1's band
    t = tp.topology_from_ini(INI_25E_LIKE.replace("mix_offset=0", "mix_offset=-700000"))
    c = t.vfos[4]
    assert c.topic == "VFO19" and c.parent == 0 and c.fs == 384000 and c.mixer_freq == float(1545116000 - (1546137300 - 700000))
    assert c.decimate_count == 3 and c.output_rate == 48000  # 384 k -> 48 k below main 1


def test_ini_rules_late_decimate_and_bufsplit():
    ini = """
sample_rate=1920000
center_frequency=1545939000
[main_vfos]
size=1
1\\frequency=1545120000
1\\out_rate=240000
[vfos]
size=2
1\\frequency=1545014429
1\\gain=4
1\\data_rate=600
1\\topic=VFO41
2\\frequency=1545045422
2\\gain=4
2\\data_rate=10500
2\\topic=VFO51
"""
    t = tp.topology_from_ini(ini)
    m, a, b = t.vfos
    assert (m.decimate_count, m.mixer_freq, t.frame) == (3, 819000.0, 480000)
    assert (a.decimate_count, a.late_decimate, a.output_rate, a.samples_per_buffer, a.mixer_freq) == (2, 5, 12000, 60000, 105571.0)
    assert (b.decimate_count, b.late_decimate, b.output_rate) == (0, 5, 48000)
    t = tp.topology_from_ini("sample_rate=288000\ncenter_frequency=1546100000\n[main_vfos]\nsize=1\n1\\frequency=1546100000\n1\\out_rate=288000\n"
                             "[vfos]\nsize=1\n1\\frequency=1546045422\n1\\gain=4\n1\\data_rate=10500\n1\\topic=VFO51\n")
    assert (t.frame, t.bufsplit) == (57600, 5)  # 2*288000/4 is not a multiple of 512
    assert (t.vfos[1].late_decimate, t.vfos[1].decimate_count, t.vfos[1].samples_per_buffer) == (6, 0, 57600)
    with pytest.raises(ValueError):
        tp.topology_from_ini("sample_rate=1000000\n")


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ROOT), reason="needs the reference's sample INIs")
def test_shipped_profiles_reproduce_embedded_tables():
    ini = open(os.path.join(REFERENCE_ROOT, "sample_ini", "sdr_25E.ini"), encoding="utf-8", errors="replace").read()
    t, e = tp.topology_from_ini(ini), tp.profile_25e()
    assert len(t.vfos) == len(e.vfos) == 29
    for a, b in zip(t.vfos, e.vfos):
        assert a == b, (a, b)
    ini = open(os.path.join(REFERENCE_ROOT, "sample_ini", "sdr_54W_all.ini"), encoding="utf-8", errors="replace").read()
    t = tp.topology_from_ini(ini)
    # SURVEY.md appendix A
    assert [(v.mixer_freq, v.decimate_count) for v in t.vfos[:3]] == [(819000.0, 3), (-181000.0, 3), (-911000.0, 3)]
    subs = t.vfos[3:]
    assert len(subs) == 14
    assert [(v.parent, v.decimate_count, v.late_decimate, v.output_rate) for v in subs[:4]] == [(0, 2, 5, 12000)] * 4
    assert [int(v.mixer_freq) for v in subs[:4]] == [105571, 90588, -14635, -74731]
    assert [(v.parent, v.decimate_count, v.late_decimate, v.output_rate, v.filter_bw) for v in subs[4:6]] == [(1, 0, 5, 48000, 0)] * 2
    assert [(v.parent, v.filter_bw) for v in subs[6:]] == [(2, 10000)] * 8
    assert [int(v.mixer_freq) for v in subs[6:]] == [32065, 26574, 21890, 16888, 11895, 7230, 1845, -3237]


def test_workload_generators():
    c3 = tp.config3()
    assert len(c3.vfos) == 1026 and len(c3.children(0)) == 512 and len(c3.children(1)) == 512
    # config 3 ~ 612 MB of algorithmic bytes per frame
    assert abs(c3.algorithmic_bytes_per_frame() / 1e6 - 612) < 2
    flat = tp.config3_flat()
    assert abs(flat.algorithmic_bytes_per_frame() / 1e9 - 3.17) < 0.01
    c4 = tp.config4()
    assert [len(c4.children(i)) for i in range(3)] == [86, 85, 85]
    assert all(v.late_decimate == 5 and v.filter_bw == 10000 for v in c4.vfos[3:])
    c2 = tp.config2()
    assert len(c2.vfos) == 2 + 32
    order = c2.leaves_in_publish_order()
    assert order[:12] == list(range(2, 14)) and len(order) == 32


def test_shard_is_a_partition():
    t = tp.config3(64)
    seen = []
    for r in range(4):
        s = tp.shard(t, r, 4)
        assert [v.parent for v in s.vfos[:2]] == [-1, -1]
        seen += [v.topic for v in s.vfos if v.parent >= 0]
        for v in s.vfos:
            assert v.parent < len(s.vfos)
    assert sorted(seen) == sorted(v.topic for v in t.vfos if v.parent >= 0)


def test_shard_never_turns_a_main_into_a_leaf():
    """More ranks than a main has subs: a main whose block on a rank is empty must not stay behind
    as a childless (compress / IQ-publishing) leaf -- the reference never publishes IQ for a main
    that has subs (vfo.cpp:253-266) -- and a rank may end up empty."""
    t = tp.profile_25e()  # 12 + 15 subs
    for world in (16, 32):
        topics = []
        for r in range(world):
            s = tp.shard(t, r, world)
            leaves = s.leaves_in_publish_order()
            assert all(s.vfos[i].demod_usb for i in leaves), (world, r)
            assert all(v.parent < 0 or s.vfos[v.parent].parent < 0 for v in s.vfos)
            topics += [s.vfos[i].topic for i in leaves]
        assert sorted(topics) == sorted(v.topic for v in t.vfos if v.parent >= 0)
    c1 = tp.config1()  # one main, one sub: rank 0 of 2 has nothing, rank 1 the whole chain
    assert [len(tp.shard(c1, r, 2).vfos) for r in range(2)] == [0, 2]


def test_shard_moves_whole_subtrees():
    """A three-level tree: the unit that moves is a sub VFO with everything below it."""
    t = tp.Topology(fs=1536000, frame=384000)
    t.vfos.append(tp.VfoDesc(parent=-1, fs=1536000, decimate_count=2, mixer_freq=484000.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=384000))
    for k in range(4):
        mid = len(t.vfos)
        t.vfos.append(tp.VfoDesc(parent=0, fs=384000, decimate_count=1, mixer_freq=1000.0 * k, demod_usb=False, cstyle=1,
                                 samples_per_buffer=96000))
        for j in range(3):
            t.vfos.append(tp.VfoDesc(topic=f"L{k}{j}", parent=mid, fs=192000, decimate_count=2, mixer_freq=500.0 * j,
                                     samples_per_buffer=48000))
    seen = []
    for r in range(2):
        s = tp.shard(t, r, 2)
        assert len(s.vfos) == 1 + 2 * 4
        leaves = s.leaves_in_publish_order()
        assert all(s.vfos[i].demod_usb for i in leaves) and len(leaves) == 6
        seen += [s.vfos[i].topic for i in leaves]
    assert sorted(seen) == sorted(v.topic for v in t.vfos if v.topic)


def test_lcg_vectorised_matches_scalar():
    x, ref = 1, []
    for _ in range(64):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        ref.append(((x >> 24) % 17) - 8)
    lcg = synth.Lcg(1)
    a = synth.lcg_frame(16, lcg)
    b = synth.lcg_frame(16, lcg)
    assert list(np.concatenate([a, b]).astype(int)) == ref
    assert a.min() >= -8 and a.max() <= 8
