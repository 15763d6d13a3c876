"""Latency structure of the compiled kernels, from the gfx950 assembly (no GPU needed): for every kernel of a source file,
how many scalar-load round trips precede the first vector memory load, which of them fetch real data (not kernel arguments),
how many vector loads are issued before the first vmcnt wait, how many loads are issued only after the first barrier
(a second memory round trip), how many vector loads are waited for (vmcnt(0)) right after being issued (one round trip
each: typically a select the compiler turned into an exec-masked load), and how many ds_bpermute (wave shuffles) are immediately waited for (serialised LDS-crossbar
round trips); plus VGPRs, spills and scratch bytes from the metadata note (an array that ends up in scratch shows here).  A dependent launch in a replayed graph costs 1.6-2.2 us on an MI355X (profiles/round2_graph_launch_floor.txt);
the small encoder/decoder kernels take 4.6-9 us, and the difference is exactly these chains.
    python tools/isa_phases.py hs_encoder.hip|/abs/path/to/source.hip [kernel-name-substring]"""
import os
import re
import subprocess
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperseg_amd import build as B


def assembly(src):
    out = os.path.join(tempfile.gettempdir(), 'hs_isa_' + os.path.basename(src).replace('.hip', '.s'))
    flags = [f for f in B.FLAGS if f != '-fPIC']
    cmd = [B._hipcc(), *flags, '-S', '--cuda-device-only', '-I', os.path.join(os.path.dirname(B.CSRC), '..', 'include'),
           '-I', B.CSRC, src if os.path.isabs(src) else os.path.join(B.CSRC, src), '-o', out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def kernels(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r'^(_Z\w+):', ln)
        if m and name is None:
            name, body = m.group(1), []
        elif name is not None:
            body.append(ln.strip())
            if ln.strip().startswith('s_endpgm'):
                yield name, body
                name = None


def resources(lines):
    """{kernel: (vgprs, spilled vgprs, scratch bytes per lane)} from the metadata note at the end of the assembly."""
    out, name, cur = {}, None, {}
    for ln in lines:
        t = ln.strip()
        if t.startswith('.name:'):
            name, cur = t.split()[-1], {}
        for key in ('.vgpr_count:', '.vgpr_spill_count:', '.private_segment_fixed_size:'):
            if name and t.startswith(key):
                cur[key] = int(t.split()[-1])
                if len(cur) == 3:
                    out[name] = (cur['.vgpr_count:'], cur['.vgpr_spill_count:'], cur['.private_segment_fixed_size:'])
    return out


def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip().split('(')[0]
    except OSError:
        return n


def analyse(body):
    is_vload = lambda s: s.startswith(('global_load', 'buffer_load', 'flat_load'))
    first_v = next((i for i, s in enumerate(body) if is_vload(s)), len(body))
    pre = body[:first_v]
    scalar_waits = sum(1 for i, s in enumerate(pre) if s.startswith('s_waitcnt') and 'lgkmcnt' in s and
                       any(p.startswith('s_load') for p in pre[max(0, i - 40):i]))
    data_sloads = sum(1 for s in pre if s.startswith('s_load') and 's[0:1]' not in s)
    first_wait = next((i for i, s in enumerate(body) if i > first_v and s.startswith('s_waitcnt') and 'vmcnt' in s), len(body))
    up_front = sum(1 for s in body[first_v:first_wait] if is_vload(s))
    total_v = sum(1 for s in body if is_vload(s))
    first_bar = next((i for i, s in enumerate(body) if s.startswith('s_barrier')), len(body))
    after_bar = sum(1 for s in body[first_bar:] if is_vload(s))
    serial_shfl = sum(1 for i, s in enumerate(body[:-2]) if s.startswith('ds_bpermute') and
                      any(t.startswith('s_waitcnt') and 'lgkmcnt(0)' in t for t in body[i + 1:i + 4]) and
                      not body[i + 1].startswith('ds_bpermute'))
    serial_loads = sum(1 for i, s in enumerate(body[:-3]) if is_vload(s) and not is_vload(body[i + 1]) and
                       any(t.startswith('s_waitcnt') and 'vmcnt(0)' in t for t in body[i + 1:i + 3]))
    return dict(serialised_loads=serial_loads, scalar_round_trips_before_first_vector_load=scalar_waits, data_s_loads_before_it=data_sloads,
                vector_loads_up_front=up_front, vector_loads=total_v, vector_loads_after_first_barrier=after_bar,
                serialised_shuffles=serial_shfl, instructions=len(body))


if __name__ == '__main__':
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    asm = assembly(sys.argv[1])
    res = resources(asm)
    for name, body in kernels(asm):
        d = demangle(name)
        if flt in d:
            a = analyse(body)
            v, sp, scr = res.get(name, (-1, -1, -1))
            print(f"{d[:90]:90s} s-trips {a['scalar_round_trips_before_first_vector_load']} (data {a['data_s_loads_before_it']})  "
                  f"vloads {a['vector_loads_up_front']}/{a['vector_loads']} up front, {a['vector_loads_after_first_barrier']} after barrier  "
                  f"serial shfl {a['serialised_shuffles']}  serial loads {a['serialised_loads']}  instr {a['instructions']}  "
                  f"vgpr {v}" + (f" SPILLS {sp}" if sp else '') + (f" scratch {scr} B" if scr else ''))
