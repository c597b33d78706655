// Token statistics of the BGZF members of a BAM file (a development probe, not product, not oracle): a plain RFC 1951 decoder that
// counts, per member, DEFLATE blocks, header symbols, literals, matches, match lengths / distances and code bits.
//   g++ -O2 -std=c++17 -o deflate_stats.bin deflate_stats.cpp && ./deflate_stats.bin file.bam [max_members]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Bits { const uint8_t* p; size_t n; uint64_t pos = 0;
	uint32_t peek(int k) { uint64_t v = 0; size_t b = pos >> 3; for (int i = 0; i < 5 && b + i < n; ++i) v |= (uint64_t)p[b + i] << (8 * i); return (uint32_t)((v >> (pos & 7)) & ((1ull << k) - 1)); }
	uint32_t get(int k) { uint32_t v = peek(k); pos += k; return v; } };

struct Huff { int cnt[16] = {0}; int sym[320]; int maxl = 0;
	void build(const int* len, int n) { memset(cnt, 0, sizeof(cnt)); for (int i = 0; i < n; ++i) cnt[len[i]]++; cnt[0] = 0; int off[16]; off[1] = 0; for (int l = 1; l < 15; ++l) off[l + 1] = off[l] + cnt[l];
		for (int i = 0; i < n; ++i) if (len[i]) sym[off[len[i]]++] = i; }
	int dec(Bits& b, int& used) { int code = 0, first = 0, idx = 0; for (int l = 1; l <= 15; ++l) { code |= (int)b.get(1); int c = cnt[l]; if (code - c < first) { used = l; return sym[idx + (code - first)]; } idx += c; first += c; first <<= 1; code <<= 1; } used = 0; return -1; } };

static const int LB[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const int LE[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const int DB[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const int DE[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

int main(int argc, char** argv)
{
	FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
	fseek(f, 0, SEEK_END); size_t sz = ftell(f); fseek(f, 0, SEEK_SET); std::vector<uint8_t> d(sz); if (fread(d.data(), 1, sz, f) != sz) return 1; fclose(f);
	long maxm = argc > 2 ? atol(argv[2]) : 1 << 30;
	uint64_t members = 0, blocks = 0, dyn = 0, fixed = 0, stored = 0, hdr_syms = 0, hdr_bits = 0, lits = 0, matches = 0, mbytes = 0, lit_bits = 0, len_bits = 0, dist_bits = 0, out_total = 0, in_total = 0;
	uint64_t lhist[259] = {0}, dhist[30] = {0}, litlen_hist[16] = {0}, nlenused[16] = {0}, selfov = 0, far1k = 0, runs = 0, runlen_hist[17] = {0};
	size_t pos = 0;
	while (pos + 18 < sz && (long)members < maxm)
	{
		uint32_t bsize = (d[pos + 16] | (d[pos + 17] << 8)) + 1; size_t cp = pos + 18, clen = bsize - 18 - 8;
		Bits b{d.data() + cp, clen}; uint64_t out = 0; int bfinal = 0; ++members; in_total += clen; uint64_t run = 0;
		do {
			bfinal = b.get(1); int bt = b.get(2); ++blocks;
			if (bt == 0) { b.pos = (b.pos + 7) & ~7ull; int n = b.get(16); b.get(16); b.pos += 8ull * n; out += n; ++stored; continue; }
			Huff L, D; int len[320] = {0};
			if (bt == 1) { ++fixed; for (int i = 0; i < 144; ++i) len[i] = 8; for (int i = 144; i < 256; ++i) len[i] = 9; for (int i = 256; i < 280; ++i) len[i] = 7; for (int i = 280; i < 288; ++i) len[i] = 8; L.build(len, 288); int dl[30]; for (int i = 0; i < 30; ++i) dl[i] = 5; D.build(dl, 30); }
			else {
				++dyn; uint64_t p0 = b.pos; int nl = b.get(5) + 257, nd = b.get(5) + 1, nc = b.get(4) + 4; static const int ord[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
				int cl[19] = {0}; for (int i = 0; i < nc; ++i) cl[ord[i]] = b.get(3); Huff C; C.build(cl, 19);
				int i = 0; while (i < nl + nd) { int u; int s = C.dec(b, u); ++hdr_syms; if (s < 16) len[i++] = s; else if (s == 16) { int r = 3 + b.get(2); int v = len[i - 1]; while (r--) len[i++] = v; } else if (s == 17) { int r = 3 + b.get(3); while (r--) len[i++] = 0; } else { int r = 11 + b.get(7); while (r--) len[i++] = 0; } }
				L.build(len, nl); D.build(len + nl, nd); hdr_bits += b.pos - p0;
				int used = 0; for (int l = 1; l < 16; ++l) used += L.cnt[l] > 0; nlenused[used]++;
			}
			for (;;)
			{
				int u; int s = L.dec(b, u); if (s < 0) { fprintf(stderr, "bad code\n"); return 2; }
				if (s < 256) { ++lits; lit_bits += u; litlen_hist[u]++; ++out; ++run; }
				else if (s == 256) break;
				else { if (run) { ++runs; runlen_hist[run > 16 ? 16 : run]++; run = 0; } int ls = s - 257; int ml = LB[ls] + b.get(LE[ls]); len_bits += u + LE[ls]; int du; int ds = D.dec(b, du); int md = DB[ds] + b.get(DE[ds]); dist_bits += du + DE[ds];
					++matches; mbytes += ml; lhist[ml]++; dhist[ds]++; if (md < ml) ++selfov; if (md > 1024) ++far1k; out += ml; }
			}
		} while (!bfinal);
		out_total += out; pos += bsize;
	}
	printf("members %llu  in %.1f B/member  out %.1f B/member  ratio %.2f\n", (unsigned long long)members, (double)in_total / members, (double)out_total / members, (double)out_total / in_total);
	printf("blocks/member %.2f (dyn %llu fixed %llu stored %llu)  header syms/block %.1f, header bits/block %.1f\n", (double)blocks / members, (unsigned long long)dyn, (unsigned long long)fixed, (unsigned long long)stored, (double)hdr_syms / (dyn ? dyn : 1), (double)hdr_bits / (dyn ? dyn : 1));
	printf("tokens/member %.0f: literals %.0f (%.1f%%, %.2f bits each), matches %.0f (mean len %.2f, len bits %.2f, dist bits %.2f)\n", (double)(lits + matches) / members, (double)lits / members, 100.0 * lits / (lits + matches), (double)lit_bits / lits, (double)matches / members, (double)mbytes / matches, (double)len_bits / matches, (double)dist_bits / matches);
	printf("bytes/token %.2f  bits/token %.2f  self-overlap %.2f%%  dist>1024 %.1f%%\n", (double)out_total / (lits + matches), 8.0 * in_total / (lits + matches), 100.0 * selfov / matches, 100.0 * far1k / matches);
	printf("literal runs/member %.0f mean run %.2f; run length hist:", (double)runs / members, (double)lits / (runs ? runs : 1)); for (int i = 1; i <= 16; ++i) printf(" %d:%.1f%%", i, 100.0 * runlen_hist[i] / (runs ? runs : 1)); printf("\n");
	printf("literal code length hist:"); for (int l = 1; l < 16; ++l) printf(" %d:%.1f%%", l, 100.0 * litlen_hist[l] / lits); printf("\n");
	printf("distinct lit/len code lengths per block:"); for (int l = 1; l < 16; ++l) if (nlenused[l]) printf(" %d:%llu", l, (unsigned long long)nlenused[l]); printf("\n");
	printf("match len hist (3..12,>12):"); uint64_t big = 0; for (int l = 13; l < 259; ++l) big += lhist[l]; for (int l = 3; l <= 12; ++l) printf(" %d:%.1f%%", l, 100.0 * lhist[l] / matches); printf(" >12:%.1f%%\n", 100.0 * big / matches);
	printf("dist sym hist:"); for (int i = 0; i < 30; ++i) printf(" %d:%.1f", i, 100.0 * dhist[i] / matches); printf("\n");
	return 0;
}
