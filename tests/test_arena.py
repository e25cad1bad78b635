"""CPU: the bookkeeping of the device memory arena (csrc/sdt_arena.h, the part of csrc/sdt_mem.hip that makes no HIP call) under a
random load of blocks taken and given back -- tools/arena_selftest.cpp checks after every few steps that no two live blocks overlap,
that live and free ranges tile every slab, that free neighbours of one slab are merged and never those of two, that the byte counts
agree and that a trim releases exactly the idle slabs.  (Every device allocation of the library goes through this book: an overlap
would be silent corruption of the node table.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def selftest(tmp_path_factory):
    exe = tmp_path_factory.mktemp("arena") / "arena_selftest"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "arena_selftest.cpp")], check=True)
    return str(exe)


@pytest.mark.parametrize("seed,steps", [(1, 20000), (2, 20000), (3, 60000)])
def test_random_load_keeps_the_book_airtight(selftest, seed, steps):
    r = subprocess.run([selftest, str(seed), str(steps)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


def test_the_library_uses_this_book():
    """sdt_mem.hip must take its ranges from sdt_arena.h (a copy of the logic inside the .hip would leave this test testing nothing)"""
    src = open(os.path.join(ROOT, "soapdenovo-trans_amd", "csrc", "sdt_mem.hip")).read()
    assert '#include "sdt_arena.h"' in src and "struct Arena : sdt::ArenaBook" in src
    assert "A.take(" in src and "A.give(" in src and "A.adopt(" in src
