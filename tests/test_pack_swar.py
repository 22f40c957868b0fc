"""k_pack_reads' 32-bases-at-a-time packer (metagraph_amd/csrc/pack_swar.hpp: SWAR over four 64-bit words) against the byte loop it
replaces (pack_read_word: KmerExtractorBOSS::encode per character, the reverse complement for strand 1) — compiled for the host
and run on 2.9 M random words with invalid, lower-case and high-bit characters.  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_swar_packer_equals_the_byte_loop(tmp_path):
    exe = str(tmp_path / "pack_swar_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "emu", "pack_swar_check.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok "), out
