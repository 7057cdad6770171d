"""Generated ROM tables: regeneration is stable, and (container only) values equal the reference's literals."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_headers_are_up_to_date_and_identical():
    a = open(os.path.join(ROOT, "oracle", "rom_tables.h")).read()
    b = open(os.path.join(ROOT, "esp32-opus-player_amd", "csrc", "rom_tables.h")).read()
    assert a == b
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rom_tables
    assert gen_rom_tables.build_text() == a


def test_tables_equal_reference_literals():
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference sources are not present on this machine")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "check_rom_tables.py")])
