"""Rewrite the generated ctypes struct blocks of INTEGRATION.md from include/hvgan.h.

    python tools/gen_integration_stub.py          # updates INTEGRATION.md in place

A block is everything between `# >>> generated: <struct>` and `# <<< generated` inside a python code fence; the text comes
from healthivert-gan_amd/lib.py:ctypes_source, i.e. from the same header parse the product's own binding uses, so the
documented binding cannot drift from the header (tests/test_host_cpu.py::test_integration_doc_struct_matches_header)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def regenerate(text):
    import hvgan  # noqa: F401
    from hvgan import lib

    def repl(m):
        return '# >>> generated: %s\n%s# <<< generated' % (m.group(1), lib.ctypes_source(m.group(1)))
    return re.sub(r'# >>> generated: (\w+)\n.*?# <<< generated', repl, text, flags=re.S)


if __name__ == '__main__':
    path = os.path.join(ROOT, 'INTEGRATION.md')
    old = open(path).read()
    new = regenerate(old)
    if new != old:
        open(path, 'w').write(new)
        print('INTEGRATION.md updated')
    else:
        print('INTEGRATION.md up to date')
