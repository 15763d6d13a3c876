"""tools/isa_phases.py reads the latency structure of a kernel off its assembly; its parser is checked on a synthetic listing
(the tool itself needs hipcc, this test does not)."""
import importlib.util
import os


def _tool():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'isa_phases.py')
    spec = importlib.util.spec_from_file_location('isa_phases', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ASM = """
_Z4demoPf:
    s_load_dwordx2 s[2:3], s[0:1], 0x0
    s_waitcnt lgkmcnt(0)
    s_load_dword s4, s[2:3], 0x0
    s_waitcnt lgkmcnt(0)
    global_load_dword v1, v[2:3], off
    global_load_dword v4, v[2:3], off offset:4
    s_waitcnt vmcnt(1)
    v_add_f32_e32 v1, v1, v1
    global_load_dword v5, v[6:7], off
    s_waitcnt vmcnt(0)
    ds_bpermute_b32 v6, v7, v1
    s_waitcnt lgkmcnt(0)
    s_barrier
    global_load_dword v8, v[6:7], off
    s_waitcnt vmcnt(0)
    global_store_dword v[2:3], v8, off
    s_endpgm
    .name:           _Z4demoPf
    .private_segment_fixed_size: 16
    .vgpr_count:     9
    .vgpr_spill_count: 2
""".strip('\n').splitlines()


def test_isa_phase_parser():
    t = _tool()
    (name, body), = list(t.kernels(ASM))
    assert name == '_Z4demoPf' and body[-1] == 's_endpgm'
    a = t.analyse(body)
    assert a['scalar_round_trips_before_first_vector_load'] == 2 and a['data_s_loads_before_it'] == 1
    assert a['vector_loads_up_front'] == 2 and a['vector_loads'] == 4 and a['vector_loads_after_first_barrier'] == 1
    assert a['serialised_loads'] == 2 and a['serialised_shuffles'] == 1
    assert t.resources(ASM) == {'_Z4demoPf': (9, 2, 16)}
