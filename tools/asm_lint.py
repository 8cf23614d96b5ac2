"""ISA lint: no instruction may read or overwrite the destination VGPRs of a vector load that is still in flight.

Vector loads return in order, so `s_waitcnt vmcnt(N)` leaves the N youngest vector-memory operations outstanding.  The compiler keeps
this invariant for its own loads; for loads issued by inline asm it cannot (it believes an asm's outputs are ready when the statement
ends): if such a register is not named by the asm wait that covers it, the compiler may copy it early or hand it out again while the load
has not landed.  That is how bag_wgrad_ws_kernel faulted when a second process stretched the latency.  The scan is
linear per kernel (branches are not followed; the queue is dropped after an unconditional branch): it is exact for the straight-line
pipelines it is meant for.

Only loads inside `;;#ASMSTART` / `;;#ASMEND` blocks are tracked by default (hipcc -S output: the compiler's own loads obey the rule by
construction, and branches make a linear scan of them noisy); --all tracks every vector load (also works on llvm-objdump -d output).

usage: python tools/asm_lint.py [--all] file.s ..."""
import re
import sys

RE_V = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
LOADS = ('global_load', 'buffer_load', 'scratch_load', 'flat_load')
STORES = ('global_store', 'buffer_store', 'scratch_store', 'flat_store', 'global_atomic', 'buffer_atomic', 'flat_atomic')


def regs(tok):
    out = set()
    for m in RE_V.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint(path, track_all=False):
    kernel, q, bad, in_asm = None, [], [], False
    for ln, line in enumerate(open(path, errors='replace'), 1):
        t = line.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        m = re.match(r'^[0-9a-f]+ <(\S+)>:$', t) or re.match(r'^(_Z\w+):', t)
        if m:
            kernel, q = m.group(1), []
            continue
        body = re.split(r'//|;', t)[0].strip()
        if not body or body.startswith('.') or body.endswith(':'):
            continue
        op = body.split()[0]
        rest = body[len(op):]
        if op in ('s_endpgm', 's_branch'):                  # nothing falls through an unconditional branch: what follows in the listing is
            q = []                                          # reached from elsewhere (an if / else pair: the persistent projection's two
            continue                                        # prologues end in different queue states) - tracking restarts there
        if op.startswith('s_waitcnt'):
            m = re.search(r'vmcnt\((\d+)\)', body)
            if m:
                del q[:max(0, len(q) - int(m.group(1)))]
            continue
        touched = regs(rest)
        hit = next((l0 for d, l0 in q if d & touched), None)
        if hit is not None:
            bad.append((kernel, ln, hit, body))
        if op.startswith(LOADS):
            q.append((regs(rest.split(',')[0]) if (in_asm or track_all) and 'lds' not in op else set(), ln))
        elif op.startswith(STORES):
            q.append((set(), ln))
    return bad


if __name__ == '__main__':
    total = 0
    track_all = '--all' in sys.argv
    for p in [a for a in sys.argv[1:] if a != '--all']:
        b = lint(p, track_all)
        total += len(b)
        print(p, len(b))
        for x in b[:8]:
            print('   ', str(x[0])[:48], 'line', x[1], 'load at line', x[2], ':', x[3][:90])
    sys.exit(1 if total else 0)
