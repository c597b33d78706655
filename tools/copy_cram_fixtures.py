#!/usr/bin/env python3
"""Copies the CRAM data files the reference's own tests hold into tests/golden/ref_in (data fixtures: inputs only; runs where /root/reference exists).
  src/cppNGS-TEST/data_in/cramTest.cram (+ .crai)              BamReader_Test.cpp:400-560 (CramSupport_* tests; their genome-independent known answers pin oracle/cram_decode.py)
  src/tools-TEST/data_in/SampleIdentity_in_{rna,wes}.cram (+ .crai)   SampleIdentity_Test.cpp:13 (the RNA file carries all its bases: RR = false)
VcfMerge.cram (1.7 MB, tools-TEST) decodes as well (43 139 records) and is left out for size."""
import os
import shutil

REF = "/root/reference"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_in")
for rel in ("src/cppNGS-TEST/data_in/cramTest.cram", "src/cppNGS-TEST/data_in/cramTest.cram.crai",
            "src/tools-TEST/data_in/SampleIdentity_in_rna.cram", "src/tools-TEST/data_in/SampleIdentity_in_rna.cram.crai",
            "src/tools-TEST/data_in/SampleIdentity_in_wes.cram", "src/tools-TEST/data_in/SampleIdentity_in_wes.cram.crai"):
    shutil.copyfile(os.path.join(REF, rel), os.path.join(DST, os.path.basename(rel)))
    print(rel)
