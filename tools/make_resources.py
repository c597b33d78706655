#!/usr/bin/env python3
"""Regenerates the DATA resources the host tools need for byte-compatible output, from the reference checkout.
Runs only where /root/reference exists (this container); the generated files are committed.
  ngs-bits_amd/resources/qcml_terms.tsv : accession<TAB>name<TAB>definition for the qcML terms the hot path emits
                                          (from src/cppNGS/Resources/qcML.obo; looked up by Statistics::addQcValue,
                                          Statistics.cpp:2904-2922)
  ngs-bits_amd/resources/hg19_snps.tsv, hg38_snps.tsv : CHROM POS REF ALT AF of the reference's known-variant resources
                                          (src/cppNGS/Resources/hg*_snps.vcf; NGSHelper::getKnownVariants)
  ngs-bits_amd/resources/qcml_tail.txt  : the fixed cvList + XSL stylesheet block every qcML file ends with
                                          (QCCollection.cpp:260-336), cut from the reference's own expected output
                                          src/tools-TEST/data_out/MappingQC_test10_out.qcML
"""
import os
import re

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "ngs-bits_amd", "resources")
WANTED = set(["QC:1000002", "QC:1000003", "QC:1000004", "QC:1000005", "QC:1000006"] +
             ["QC:20000%02d" % i for i in list(range(19, 33)) + [37, 38, 50, 51, 52, 57, 58, 59, 60, 61] + list(range(65, 77)) + list(range(90, 99)) + [99]] +
             ["QC:20000%02d" % i for i in range(5, 13)] + ["QC:2000131", "QC:2000132", "QC:2000138"] + ["QC:20001%02d" % i for i in range(43, 49)] + ["QC:2000139", "QC:2000150"] + ["QC:20001%02d" % i for i in range(0, 10)])


def main():
    obo = open(os.path.join(REF, "src/cppNGS/Resources/qcML.obo"), encoding="utf-8").read()
    rows = []
    for term in obo.split("[Term]")[1:]:
        tid = re.search(r"^id: (\S+)", term, re.M)
        name = re.search(r"^name: (.*)$", term, re.M)
        d = re.search(r'^def: "(.*?)(?<!\\)"', term, re.M)
        if tid and tid.group(1) in WANTED:
            rows.append((tid.group(1), name.group(1).strip(), d.group(1) if d else ""))
    rows.sort()
    with open(os.path.join(RES, "qcml_terms.tsv"), "w", encoding="utf-8") as f:
        for r in rows:
            f.write("\t".join(r) + "\n")
    exp = open(os.path.join(REF, "src/tools-TEST/data_out/MappingQC_test10_out.qcML"), encoding="utf-8").read()
    tail = exp[exp.index("  <cvList>"):]
    with open(os.path.join(RES, "qcml_tail.txt"), "w", encoding="utf-8") as f:
        f.write(tail)
    print(len(rows), "terms;", len(tail), "bytes of tail")
    # known common variants of NGSHelper::getKnownVariants (resources hg19_snps.vcf / hg38_snps.vcf): the five columns the
    # contamination check reads (CHROM, POS, REF, ALT, INFO/AF), one line per VCF record, order kept
    for build in ("hg19", "hg38"):
        n = 0
        with open(os.path.join(RES, f"{build}_snps.tsv"), "w") as f:
            for ln in open(os.path.join(REF, f"src/cppNGS/Resources/{build}_snps.vcf")):
                if ln.startswith("#"):
                    continue
                c = ln.rstrip("\n").split("\t")
                af = ""
                for kv in c[7].split(";"):
                    if kv.startswith("AF="):
                        af = kv[3:]
                f.write("\t".join([c[0], c[1], c[3], c[4], af]) + "\n"); n += 1
        print(build, n, "known variants")


if __name__ == "__main__":
    main()
