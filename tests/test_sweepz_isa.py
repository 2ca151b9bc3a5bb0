"""csrc/xee_sweepz.hip reads the next tick's row from LDS a tick ahead with inline assembly and waits for it at the tick's
end; in between, the destination registers are pending and the hardware does not interlock them.  The compiler does not
know: this test compiles the kernel (hipcc cross-compiles gfx950 without a GPU) and scans the assembly for any instruction
that touches a pending register before the next lgkmcnt wait (tools/isa_lds_hazards.py)."""
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


@pytest.mark.skipif(not shutil.which("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_instruction_touches_a_pending_lds_destination():
    import isa_lds_hazards as T
    reads, hazards = T.scan(T.compile_asm())
    assert reads >= 32, reads                      # the issue statements are there (4 reads x 8 ticks x the optimistic walk, + the exact walk's)
    assert not hazards, hazards[:5]


def test_the_scanner_sees_a_hazard(tmp_path):
    import isa_lds_hazards as T
    bad = tmp_path / "bad.s"
    bad.write_text("\tds_read2st64_b32 v[4:5], v9 offset0:1 offset1:3\n\tv_mov_b64_e32 v[10:11], v[4:5]\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 v[0:1], v[4:5], v[2:3]\n")
    reads, hazards = T.scan(bad)
    assert reads == 1 and len(hazards) == 2 and hazards[0][0] == 2
    ok = tmp_path / "ok.s"
    ok.write_text("\tds_read2st64_b32 v[4:5], v9 offset0:1 offset1:3\n\tv_pk_mul_f32 v[6:7], v[0:1], v[2:3]\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 v[0:1], v[4:5], v[2:3]\n")
    assert T.scan(ok) == (1, [])
