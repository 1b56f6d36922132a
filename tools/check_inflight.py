"""Build-time check of the loss kernels' hand-pipelined scalar loads.

k_loss_main issues the scalar loads of the next turn's ground-truth rows in one inline-asm statement
(turn_issue) and waits for them in another (turn_wait).  Between the two the registers being loaded must
not be touched: the compiler knows nothing of the loads in flight, and a copy, a spill (v_writelane) or a
reuse of one of those registers there would capture or destroy a value that has not arrived yet.  The
constraints of the asm statements make that unlikely, not impossible, so the generated code is checked:

    python tools/check_inflight.py <assembly of yh_loss.hip (hipcc -S --cuda-device-only)>

For every kernel, from the end of every asm block that holds s_load instructions, every path of the
control-flow graph is followed up to the next asm block that holds `s_waitcnt lgkmcnt(0)`; an instruction
on the way that names one of the block's destination registers is an error.  Exit status 1 on any."""
import re
import sys

REG = re.compile(r'\bs\[(\d+):(\d+)\]|\bs(\d+)\b')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def functions(lines):
    name, body = None, []
    for ln in lines:
        if re.match(r'^[A-Za-z_][\w$.]*:\s*(;.*)?$', ln) and not ln.startswith('.L'):
            name, body = ln.split(':')[0], []
        elif ln.startswith('.Lfunc_end') and name:
            yield name, body
            name = None
        elif name is not None:
            body.append(ln)


def check(name, body):
    # instructions with their asm-block membership; labels -> instruction index
    ins, labels, in_asm, block_id = [], {}, False, 0
    for ln in body:
        t = ln.strip()
        if not t or t.startswith(';') and 'ASM' not in t:
            continue
        if t.startswith(';;#ASMSTART'):
            in_asm, block_id = True, block_id + 1
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        m = re.match(r'^(\.L\w+):', t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith('.'):
            continue
        ins.append((t.split(';')[0].strip(), block_id if in_asm else 0))
    errors = []
    issue_blocks = sorted({b for t, b in ins if b and t.startswith('s_load')})
    for blk in issue_blocks:
        idx = [i for i, (t, b) in enumerate(ins) if b == blk]
        dest = set()
        for i in idx:
            t = ins[i][0]
            if t.startswith('s_load'):
                dest |= regs_of(t.split(',')[0])
        seen, stack = set(), [idx[-1] + 1]
        while stack:
            i = stack.pop()
            while i < len(ins) and i not in seen:
                seen.add(i)
                t, b = ins[i]
                if b and 's_waitcnt lgkmcnt(0)' in t:
                    break                                   # the wait: this path is done
                touched = regs_of(t) & dest
                if touched:
                    errors.append('%s: `%s` touches s%s while the loads of asm block %d are in flight'
                                  % (name, t, sorted(touched), blk))
                op = t.split()[0]
                if op == 's_endpgm':
                    break
                if op == 's_branch':
                    i = labels[t.split()[1]]
                    continue
                if op.startswith('s_cbranch'):
                    stack.append(labels[t.split()[1]])
                i += 1
    return len(issue_blocks), errors


def main():
    lines = open(sys.argv[1]).read().split('\n')
    total, bad = 0, []
    for name, body in functions(lines):
        if 'k_loss_main' not in name:
            continue
        n, errors = check(name, body)
        total += n
        bad += errors
        print('%-70s %3d asm blocks with loads in flight, %d violations' % (name[:70], n, len(errors)))
    for e in bad[:40]:
        print('  ' + e)
    if total == 0:
        print('no pipelined scalar loads found: the check has nothing to look at')
        return 1
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
