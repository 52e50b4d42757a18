"""Lists the kernels of csrc/libp3d_hip.so whose gfx950 code contains packed-fp32 VALU instructions (see csrc/Makefile for why there must be none)."""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, '3d-pose-estimation-with-previleged-information_amd', 'csrc', 'libp3d_hip.so')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
PACKED = re.compile(r'\bv_pk_(add|mul|fma)_f32\b')


def scan(so=SO):
    """{kernel: count} of packed-fp32 instructions, and the number of kernels looked at: every ELF64 code object embedded in the fat binary is disassembled"""
    data = open(so, 'rb').read()
    found, kernels = {}, 0
    with tempfile.TemporaryDirectory() as tmp:
        for n, m in enumerate(re.finditer(b'\x7fELF', data)):
            i = m.start()
            if i == 0 or data[i + 4] != 2:
                continue
            e_shoff = struct.unpack_from('<Q', data, i + 0x28)[0]
            e_shentsize, e_shnum = struct.unpack_from('<HH', data, i + 0x3A)
            path = os.path.join(tmp, 'co%d.o' % n)
            with open(path, 'wb') as f:
                f.write(data[i:i + e_shoff + e_shentsize * e_shnum])
            out = subprocess.run([OBJDUMP, '-d', path], capture_output=True, text=True).stdout
            cur = None
            for line in out.splitlines():
                mm = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
                if mm:
                    cur = mm.group(1)
                    kernels += 1
                elif PACKED.search(line):
                    found[cur] = found.get(cur, 0) + 1
    return found, kernels


if __name__ == '__main__':
    found, kernels = scan()
    for k, v in sorted(found.items(), key=lambda kv: -kv[1]):
        print(v, k)
    print('%d of %d kernels contain packed-fp32 instructions' % (len(found), kernels))
    sys.exit(1 if found else 0)
