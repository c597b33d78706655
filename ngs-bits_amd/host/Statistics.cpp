#include <exception>
#include <mutex>
#include <atomic>
#include "Statistics.hpp"
#include <ctime>
#include <thread>
#include <zlib.h>

namespace ngsbits {

const char* const NO_REF = "<none>";

// ---------------------------------------------------------------- BamReader
BamReader::BamReader(const std::string& bam_file, const std::string& ref, bool allow_shards) : bam_file_(bam_file) { init(ref, allow_shards, nullptr, 0); }
BamReader::BamReader(const std::string& bam_file, const std::string& ref, bool allow_shards, const BedFile& regions) : bam_file_(bam_file) { init(ref, allow_shards, &regions, 0); }
BamReader::BamReader(const std::string& bam_file, const std::string& ref, Head head) : bam_file_(bam_file) { init(ref, false, nullptr, head.n_members); }
static double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static const double g_t0 = now_s();
static void stamp(const char* what) { if (getenv("NGSQC_TIMING")) fprintf(stderr, "[ngsqc] +%.3f s %s\n", now_s() - g_t0, what); }

void BamReader::requireTags(bool needed) { ngsqc_set_cram_skip(NGSQC_CRAM_SKIP_NAMES | (needed ? 0 : NGSQC_CRAM_SKIP_TAGS)); }
static const bool g_cram_skip_default = (BamReader::requireTags(false), true);   // (tools: names never, tags only where a function says so)

void BamReader::init(const std::string& ref, bool allow_shards, const BedFile* regions, int64_t head_members)
{
	head_members_ = head_members;
	stamp("open: start");
	ref_file_ = ref;
	// CRAM input (BamReader.cpp:482-492: hts_set_fai_filename with the genome the caller names, else the one of the settings): the library decodes against it
	if (ref != NO_REF) { std::string r = ref.empty() ? defaultReferenceGenome() : ref; ngsqc_set_reference(r.empty() ? nullptr : r.c_str()); }
	int dev = 0; if (const char* e = getenv("NGSQC_DEVICE")) dev = atoi(e);
	int n_shards = 1; if (allow_shards) if (const char* e = getenv("NGSQC_SHARDS")) n_shards = std::max(1, atoi(e));
	int n_dev = 1; if (n_shards > 1) if (const char* e = getenv("NGSQC_DEVICES")) n_dev = std::max(1, atoi(e));
	const char* es = getenv("NGSQC_INDEX_SELECT");
	const bool by_index = regions && n_shards == 1 && (!es || atoi(es) != 0) && hasBamIndex(bam_file_);
	for (int s = 0; s < n_shards; ++s)
	{
		ngsqc_handle* h = nullptr; int rc;
		if (head_members > 0) rc = ngsqc_open_head(bam_file_.c_str(), dev, head_members, &h);
		else if (by_index)
		{
			std::vector<std::string> names; std::vector<ngsqc_named_region> nr;
			names.reserve((size_t)regions->count());
			for (long long i = 0; i < regions->count(); ++i) names.push_back((*regions)[i].chr().str());
			for (long long i = 0; i < regions->count(); ++i) nr.push_back(ngsqc_named_region{names[(size_t)i].c_str(), (*regions)[i].start(), (*regions)[i].end()});
			rc = ngsqc_open_regions(bam_file_.c_str(), dev, nr.data(), (int64_t)nr.size(), &h);
		}
		else rc = n_shards == 1 ? ngsqc_open(bam_file_.c_str(), dev, &h) : ngsqc_open_shard(bam_file_.c_str(), dev + s % n_dev, s, n_shards, &h);
		if (rc != NGSQC_OK)
		{
			std::string msg = ngsqc_last_error(nullptr);
			for (ngsqc_handle* o : shards_) ngsqc_close(o);
			shards_.clear();
			if (rc == NGSQC_E_IO && msg.find("Could not load index") != std::string::npos) NB_THROW(FileAccessException, msg);
			if (rc == NGSQC_E_IO && msg.find("Error while setting reference genome") != std::string::npos) NB_THROW(FileAccessException, msg);   // BamReader.cpp:486-489 (CRAM)
			if (rc == NGSQC_E_IO) NB_THROW(FileAccessException, "Could not open BAM/CRAM file " + bam_file_);   // BamReader.cpp:467
			if (rc == NGSQC_E_DEVICE) NB_THROW(Exception, "GPU backend unavailable: " + msg);
			NB_THROW(FileAccessException, msg);
		}
		shards_.push_back(h);
	}
	h_ = shards_[0];
	for (int i = 0; i < ngsqc_n_ref(h_); ++i) { chrs_.emplace_back(ngsqc_ref_name(h_, i)); sizes_.push_back(ngsqc_ref_len(h_, i)); }
	stamp("open: done");
	if (getenv("NGSQC_TIMING") && (by_index || head_members > 0)) fprintf(stderr, "[ngsqc] %s: %lld BGZF members on the device\n", by_index ? "index-driven open" : "head open", (long long)ngsqc_n_bgzf_blocks(h_));
}
// BamReader::info (BamReader.cpp:593-730): format, genome build, mapper from the last @PG line, paired-end from the first 100 usable reads, the hg38
// false-duplication mask from a region query, alt contigs from the header
BamInfo BamReader::info()
{
	BamInfo out;
	out.file_format = "BAM";
	{   // CRAM container version (BamReader.cpp:603-617)
		std::ifstream f(bam_file_, std::ios::binary); char hd[6] = {0, 0, 0, 0, 0, 0};
		if (f.read(hd, 6) && memcmp(hd, "CRAM", 4) == 0) out.file_format = "CRAM " + std::to_string((int)(uint8_t)hd[4]) + "." + std::to_string((int)(uint8_t)hd[5]);
	}
	try { const int c1 = chromosomeSize(Chromosome("chr1")); out.build = c1 == 249250621 ? "hg19" : (c1 == 248956422 ? "hg38" : ""); } catch (...) {}
	// paired end: the first 100 reads OF THE FILE that are not secondary / supplementary / duplicate / unmapped and have MAPQ >= 20 (BamReader.cpp:627-640 keeps
	// reading until it has them). A head open only holds the first members: when those do not yield 100 such reads (MAPQ < 20 at the telomere repeats of a WGS
	// BAM, unmapped or duplicate reads in front) the head is opened again four times as long, until it does or the file has no more records to offer
	{
		auto count = [](ngsqc_handle* hh, BamReader* rd, double& n_all, double& n_paired) -> int64_t {
			const int64_t nbytes = ngsqc_inflated_size(hh), nrec = ngsqc_n_records(hh);
			if (nrec < 0) rd->check((int)nrec);   // (an error code: not "no reads")
			std::vector<uint8_t> infl((size_t)std::max<int64_t>(nbytes, 1)); std::vector<int64_t> off((size_t)std::max<int64_t>(nrec, 1));
			if (nrec > 0) { rd->check(ngsqc_copy_inflated(hh, infl.data(), nbytes)); rd->check(ngsqc_copy_record_offsets(hh, off.data(), nrec)); }
			n_all = 0; n_paired = 0;
			for (int64_t i = 0; i < nrec && n_all < 100.0; ++i)
			{
				const uint8_t* r = infl.data() + off[(size_t)i];
				const uint32_t w = (uint32_t)r[12] | ((uint32_t)r[13] << 8), flag = (uint32_t)r[18] | ((uint32_t)r[19] << 8), mapq = w >> 8;
				if (flag & (0x100 | 0x800 | 0x400 | 0x4)) continue;
				if (mapq < 20) continue;
				if (flag & 0x1) n_paired += 1.0;
				n_all += 1.0;
			}
			return nrec;
		};
		double n_all = 0, n_paired = 0;
		int64_t nrec = count(h_, this, n_all, n_paired), head = head_members_;
		while (n_all < 100.0 && head > 0)
		{
			head = head >= (1ll << 40) ? 0 : head * 4;
			BamReader more(bam_file_, ref_file_, Head{head > 0 ? head : (1ll << 62)});
			double a = 0, p = 0; const int64_t n2 = count(more.handle(), &more, a, p);
			n_all = a; n_paired = p;
			if (n2 <= nrec) break;   // the longer head brought no further record: that was the whole file
			nrec = n2;
		}
		out.paired_end = n_paired / n_all > 0.1;   // (0/0 = nan: false, like the reference)
	}
	// mapper: the last @PG line counts
	{
		std::string text((size_t)ngsqc_header_text(h_, nullptr, 0) + 1, '\0');
		ngsqc_header_text(h_, &text[0], (int64_t)text.size()); text.resize(text.size() - 1);
		std::vector<std::string> lines; size_t a = 0;
		while (a <= text.size()) { size_t b = text.find('\n', a); if (b == std::string::npos) b = text.size(); if (b > a) lines.push_back(text.substr(a, b - a)); a = b + 1; }
		auto vn = [](const std::string& line) { std::string v; size_t a = 0; while (a <= line.size()) { size_t b = line.find('\t', a); if (b == std::string::npos) b = line.size(); if (line.compare(a, 3, "VN:") == 0) v = trimmed(line.substr(a + 3, b - a - 3)); a = b + 1; } return v; };
		for (size_t i = lines.size(); i-- > 0;)
		{
			const std::string& line = lines[i];
			if (line.compare(0, 3, "@PG") != 0) continue;
			if (line.find("PN:bwa-mem2") != std::string::npos) { out.mapper = "bwa-mem2"; out.mapper_version = vn(line); break; }
			if (line.find("PN:bwa") != std::string::npos) { out.mapper = "bwa"; out.mapper_version = vn(line); break; }
			if (line.find("ID: DRAGEN SW build") != std::string::npos)
			{
				out.mapper = "DRAGEN"; std::string v = vn(line); std::vector<std::string> parts; size_t a2 = 0;
				while (a2 <= v.size()) { size_t b = v.find('.', a2); if (b == std::string::npos) b = v.size(); parts.push_back(v.substr(a2, b - a2)); a2 = b + 1; }
				std::string ver; for (size_t k = parts.size() > 3 ? parts.size() - 3 : 0; k < parts.size(); ++k) ver += (ver.empty() ? "" : ".") + parts[k];
				if (!v.empty()) out.mapper_version = ver;
				break;
			}
			if (line.find("PN:minimap2") != std::string::npos) { out.mapper = "minimap2"; out.mapper_version = vn(line); break; }
			if (line.find("PN:STAR") != std::string::npos) { out.mapper = "STAR"; std::string v = vn(line); size_t k; while ((k = v.find("STAR_")) != std::string::npos) v.erase(k, 5); out.mapper_version = v; break; }
		}
	}
	// hg38: a read in the region that the false-duplication mask empties means the genome was not masked (region query through the index)
	if (out.build == "hg38")
	{
		try
		{
			BedFile roi; roi.append(BedLine(Chromosome("chr21"), 5968000, 6160000));
			BamReader q(bam_file_, ref_file_, false, roi);
			q.requireIndex();
			const int tid = q.chromosomeID(Chromosome("chr21"));
			const int64_t nbytes = ngsqc_inflated_size(q.handle()), nrec = ngsqc_n_records(q.handle());
			if (tid >= 0 && nrec > 0)
			{
				std::vector<uint8_t> infl((size_t)nbytes); std::vector<int64_t> off((size_t)nrec);
				q.check(ngsqc_copy_inflated(q.handle(), infl.data(), nbytes)); q.check(ngsqc_copy_record_offsets(q.handle(), off.data(), nrec));
				for (int64_t i = 0; i < nrec; ++i)
				{
					const uint8_t* r = infl.data() + off[(size_t)i];
					auto rd = [&](int o) { return (int32_t)((uint32_t)r[o] | ((uint32_t)r[o + 1] << 8) | ((uint32_t)r[o + 2] << 16) | ((uint32_t)r[o + 3] << 24)); };
					const int32_t rt = rd(4), pos = rd(8); const uint32_t l_name = r[12], n_cig = (uint32_t)r[16] | ((uint32_t)r[17] << 8), flag = (uint32_t)r[18] | ((uint32_t)r[19] << 8);
					long long ref_len = 0;
					for (uint32_t k = 0; k < n_cig; ++k) { const uint32_t c = (uint32_t)rd(36 + (int)l_name + 4 * (int)k); if ((0x18Du >> (c & 15u)) & 1u) ref_len += c >> 4; }
					const long long end = pos + ((flag & 0x4) || ref_len == 0 ? 1 : ref_len);   // bam_endpos
					if (rt == tid && pos < 6160000 && end > 5968000 - 1) { out.false_duplications_masked = false; break; }   // the iterator's overlap rule for chr21:5968000-6160000
				}
			}
		}
		catch (...) {}   // (the reference swallows the exception of an empty range / a missing index here)
	}
	for (const Chromosome& c : chrs_)
	{
		std::string name = c.str(); for (auto& ch : name) ch = (char)tolower((unsigned char)ch);
		auto ends = [&](const char* suf) { const size_t k = strlen(suf); return name.size() >= k && name.compare(name.size() - k, k, suf) == 0; };
		if (ends("_alt") || ends("_hap1")) { out.contains_alt_chrs = true; break; }
	}
	return out;
}

BamReader::~BamReader() { stamp("close: start"); for (ngsqc_handle* h : shards_) ngsqc_close(h); stamp("close: done"); }
int BamReader::chromosomeID(const Chromosome& chr) const { for (size_t i = 0; i < chrs_.size(); ++i) if (chrs_[i] == chr) return (int)i; return -1; }
int BamReader::chromosomeSize(const Chromosome& chr) const
{
	int id = chromosomeID(chr);
	if (id < 0) NB_THROW(ArgumentException, "Chromosome '" + chr.str() + "' not known in BAM/CRAM file " + bam_file_);
	return (int)sizes_[(size_t)id];
}
double BamReader::genomeSize(bool include_special) const { double s = 0; for (size_t i = 0; i < chrs_.size(); ++i) if (chrs_[i].isNonSpecial() || include_special) s += (double)sizes_[i]; return s; }
void BamReader::requireIndex() const
{
	// the reference needs a .bai/.csi for every region query; the GPU path does not, but the error is part of the contract
	if (hasBamIndex(bam_file_)) return;
	NB_THROW(FileAccessException, "Could not load index of BAM/CRAM file " + bam_file_);
}
void BamReader::check(int rc) const
{
	if (rc == NGSQC_OK) return;
	std::string msg = ngsqc_last_error(h_);
	if (rc == NGSQC_E_ARG) NB_THROW(ArgumentException, msg);
	if (rc == NGSQC_E_FORMAT) NB_THROW(FileAccessException, msg);
	NB_THROW(Exception, msg);
}

// ---------------------------------------------------------------- QC value helpers
void Statistics::addQcValue(QCCollection& output, const std::string& accession, const std::string& name, double value)
{
	auto it = qcmlTerms().find(accession);
	if (it == qcmlTerms().end()) NB_THROW(ProgrammingException, "qcML does not contain term with accession '" + accession + "'!");
	if (it->second.name != name) NB_THROW(ProgrammingException, "qcML term with accession '" + accession + "' does not have name '" + name + "'!");
	QCValue v; v.name = name; v.accession = accession; v.description = it->second.definition; v.type = QCValueType::DOUBLE; v.d = value; output.insert(v);
}
void Statistics::addQcValue(QCCollection& output, const std::string& accession, const std::string& name, const std::string& value)
{
	auto it = qcmlTerms().find(accession);
	if (it == qcmlTerms().end()) NB_THROW(ProgrammingException, "qcML does not contain term with accession '" + accession + "'!");
	if (it->second.name != name) NB_THROW(ProgrammingException, "qcML term with accession '" + accession + "' does not have name '" + name + "'!");
	QCValue v; v.name = name; v.accession = accession; v.description = it->second.definition; v.type = QCValueType::STRING; v.s = value; output.insert(v);
}

// Minimal line plot -> PNG -> base64 (the reference renders with QtCharts; payloads are not compared by its tests,
// MappingQC_Test.cpp:15-16). Axes + polylines, no text.
static std::string base64(const std::vector<uint8_t>& d)
{
	static const char* T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/"; std::string o;
	for (size_t i = 0; i < d.size(); i += 3)
	{
		uint32_t v = d[i] << 16 | (i + 1 < d.size() ? d[i + 1] << 8 : 0) | (i + 2 < d.size() ? d[i + 2] : 0);
		o += T[(v >> 18) & 63]; o += T[(v >> 12) & 63]; o += i + 1 < d.size() ? T[(v >> 6) & 63] : '='; o += i + 2 < d.size() ? T[v & 63] : '=';
	}
	return o;
}
static std::string plotPng(const std::vector<double>& x, const std::vector<std::vector<double>>& lines)
{
	const int W = 640, H = 480, L = 40, B = 30; std::vector<uint8_t> img((size_t)W * H * 3, 255);
	auto px = [&](int xx, int yy, uint8_t r, uint8_t g, uint8_t b) { if (xx < 0 || yy < 0 || xx >= W || yy >= H) return; size_t o = ((size_t)yy * W + xx) * 3; img[o] = r; img[o + 1] = g; img[o + 2] = b; };
	for (int i = L; i < W - 10; ++i) px(i, H - B, 0, 0, 0);
	for (int j = 10; j <= H - B; ++j) px(L, j, 0, 0, 0);
	double xmin = 1e300, xmax = -1e300, ymin = 0, ymax = -1e300;
	for (double v : x) { xmin = std::min(xmin, v); xmax = std::max(xmax, v); }
	for (auto& l : lines) for (double v : l) if (std::isfinite(v)) ymax = std::max(ymax, v);
	if (!(xmax > xmin)) xmax = xmin + 1;
	if (!(ymax > ymin)) ymax = ymin + 1;
	const uint8_t cols[3][3] = {{31, 119, 180}, {255, 127, 14}, {44, 160, 44}}; int li = 0;
	for (auto& l : lines)
	{
		int lx = -1, ly = -1;
		for (size_t i = 0; i < l.size() && i < x.size(); ++i)
		{
			if (!std::isfinite(l[i])) continue;
			int cx = L + (int)((x[i] - xmin) / (xmax - xmin) * (W - L - 11)), cy = H - B - (int)((l[i] - ymin) / (ymax - ymin) * (H - B - 11));
			if (lx >= 0) { int n = std::max(abs(cx - lx), abs(cy - ly)) + 1; for (int k = 0; k <= n; ++k) px(lx + (cx - lx) * k / n, ly + (cy - ly) * k / n, cols[li % 3][0], cols[li % 3][1], cols[li % 3][2]); }
			lx = cx; ly = cy;
		}
		++li;
	}
	std::vector<uint8_t> raw; raw.reserve((size_t)(W * 3 + 1) * H);
	for (int j = 0; j < H; ++j) { raw.push_back(0); raw.insert(raw.end(), img.begin() + (size_t)j * W * 3, img.begin() + (size_t)(j + 1) * W * 3); }
	uLongf clen = compressBound((uLong)raw.size()); std::vector<uint8_t> comp(clen); compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6); comp.resize(clen);
	std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
	auto be32 = [](std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back((x >> 16) & 255); v.push_back((x >> 8) & 255); v.push_back(x & 255); };
	auto chunk = [&](const char* type, const std::vector<uint8_t>& data) { be32(png, (uint32_t)data.size()); std::vector<uint8_t> td(type, type + 4); td.insert(td.end(), data.begin(), data.end()); png.insert(png.end(), td.begin(), td.end()); be32(png, (uint32_t)crc32(0, td.data(), (uInt)td.size())); };
	std::vector<uint8_t> ihdr; be32(ihdr, W); be32(ihdr, H); ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
	chunk("IHDR", ihdr); chunk("IDAT", comp); chunk("IEND", {});
	return base64(png);
}
void Statistics::addQcPlot(QCCollection& output, const std::string& accession, const std::string& name, const std::vector<double>& x, const std::vector<std::vector<double>>& lines)
{
	auto it = qcmlTerms().find(accession);
	if (it == qcmlTerms().end()) NB_THROW(ProgrammingException, "qcML does not contain term with accession '" + accession + "'!");
	if (it->second.name != name) NB_THROW(ProgrammingException, "qcML term with accession '" + accession + "' does not have name '" + name + "'!");
	QCValue v; v.name = name; v.accession = accession; v.description = it->second.definition; v.type = QCValueType::IMAGE; v.s = plotPng(x, lines); output.insert(v);
}

// ---------------------------------------------------------------- shared pieces of the three mapping variants
namespace {

struct GcPrep
{
	BedFile dropout; std::vector<int32_t> bin; std::vector<double> gc_roi = std::vector<double>(101, 0.0); bool enabled = false;
	// Statistics.cpp:363-387 / 1022-1045
	GcPrep(const BedFile& roi, FastaFileIndex* fa)
	{
		if (!fa) return;
		enabled = true; dropout.add(roi); dropout.chunk(100); bin.assign((size_t)dropout.count(), -1);
		for (long long i = 0; i < dropout.count(); ++i)
		{
			const BedLine& l = dropout[i];
			double gc = gcContent(fa->seq(l.chr(), l.start(), l.length()));
			if (!std::isfinite(gc)) continue;
			int b = (int)std::floor(100.0 * gc); bin[(size_t)i] = b; gc_roi[(size_t)std::min(b, 100)] += 1.0;
		}
	}
};

std::vector<ngsqc_region> toRegions(const BedFile& bed, const BamReader& reader, bool need_all)
{
	std::vector<ngsqc_region> r;
	for (long long i = 0; i < bed.count(); ++i)
	{
		int tid = reader.chromosomeID(bed[i].chr());
		if (tid < 0) { if (need_all) NB_THROW(FileAccessException, "Could not find chromosome '" + bed[i].chr().str() + "' in BAM/CRAM file " + reader.fileName()); continue; } // BamReader.cpp:751-754
		r.push_back(ngsqc_region{tid, bed[i].start(), bed[i].end()});
	}
	return r;
}

struct KnownSnp { Chromosome chr; int pos; char ref, alt; };
struct SnpSites { std::vector<KnownSnp> snps; std::vector<ngsqc_region> sites; std::vector<size_t> slot; };   // slot[i]: row of snps[i] in the (tid, pos)-sorted site table

std::vector<KnownSnp> loadKnownSnps(const std::string& build, const std::string& roi_file, const BedFile* roi_in = nullptr)
{
	// target region: variants whose [pos, pos + len(ref) - 1] overlaps a line are kept (VcfFile::setRegion / VcfFile.cpp:131-136)
	std::map<int, std::vector<std::pair<int, int>>> roi_by_chr; std::map<int, std::vector<int>> roi_pmax;
	const bool have_roi = roi_file != "" || roi_in != nullptr;
	if (have_roi)
	{
		BedFile roi; if (roi_in) roi.add(*roi_in); else roi.load(roi_file);
		roi.sort();
		for (long long i = 0; i < roi.count(); ++i) roi_by_chr[roi[i].chr().num()].push_back({roi[i].start(), roi[i].end()});
		for (auto& kv : roi_by_chr) { std::vector<int>& pm = roi_pmax[kv.first]; int m = 0; for (auto& se : kv.second) { m = std::max(m, se.second); pm.push_back(m); } }
	}
	auto in_roi = [&](const Chromosome& chr, int s, int e) {
		auto it = roi_by_chr.find(chr.num()); if (it == roi_by_chr.end()) return false;
		const auto& v = it->second; const auto& pm = roi_pmax[chr.num()];
		size_t hi = (size_t)(std::upper_bound(v.begin(), v.end(), std::make_pair(e, std::numeric_limits<int>::max())) - v.begin());   // lines with start <= e
		return hi > 0 && pm[hi - 1] >= s;
	};
	// known variants: SNVs with 0.2 <= AF <= 0.8 (getKnownVariants(build, true, [roi,] 0.2, 0.8))
	std::string res = resourceDir() + "/" + build + "_snps.tsv";
	std::ifstream f(res);
	if (!f) NB_THROW(ProgrammingException, "Unsupported genome build '" + build + "'!");   // NGSHelper.cpp copyFromResource
	std::vector<KnownSnp> snps; std::string line;
	while (std::getline(f, line))
	{
		std::vector<std::string> c = split(line, '\t');
		if (c.size() < 5) continue;
		const int pos = atoi(c[1].c_str());
		Chromosome chr(c[0]);
		if (have_roi && !in_roi(chr, pos, pos + (int)c[2].size() - 1)) continue;
		char* end = nullptr; double af = c[4].empty() ? 0.0 : strtod(c[4].c_str(), &end); if (c[4].empty() || *end) af = 0.0;   // QByteArray::toDouble
		if (!(af >= 0.2 && af <= 0.8)) continue;
		std::string alt0 = c[3].substr(0, c[3].find(','));
		for (auto& ch : alt0) ch = (char)toupper(ch);
		if (!(alt0.size() == 1 && c[2].size() == 1 && alt0 != "-" && c[2] != "-")) continue;                                    // VcfLine::isSNV
		snps.push_back(KnownSnp{chr, pos, c[2][0], alt0[0]});
	}
	return snps;
}

// site table for the GPU: (tid, pos), grouped by tid in position order
SnpSites snpSites(const BamReader& reader, std::vector<KnownSnp> snps)
{
	SnpSites t; t.snps.swap(snps);
	if (!t.snps.empty()) reader.requireIndex();                                                                                // getPileup -> setRegion (BamReader.cpp:740-746)
	std::vector<int> tids(t.snps.size()); std::vector<size_t> order;
	for (size_t i = 0; i < t.snps.size(); ++i)
	{
		tids[i] = reader.chromosomeID(t.snps[i].chr);
		if (tids[i] < 0) NB_THROW(FileAccessException, "Could not find chromosome '" + t.snps[i].chr.str() + "' in BAM/CRAM file " + reader.fileName());   // BamReader.cpp:750-754
		order.push_back(i);
	}
	std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return tids[a] != tids[b] ? tids[a] < tids[b] : t.snps[a].pos < t.snps[b].pos; });
	for (size_t k : order) t.sites.push_back(ngsqc_region{tids[k], t.snps[k].pos, t.snps[k].pos});
	t.slot.resize(t.snps.size()); for (size_t j = 0; j < order.size(); ++j) t.slot[order[j]] = j;
	return t;
}

QCCollection contaminationFromCounts(const SnpSites& t, const std::vector<int64_t>& counts, bool debug, int min_cov, int min_snps)
{
	const std::vector<KnownSnp>& snps = t.snps;
	Histogram hist(0, 1, 0.05);
	int passed = 0; double passed_depth_sum = 0.0;
	for (size_t i = 0; i < snps.size(); ++i)                                                                                   // file order, as the reference
	{
		const int64_t* c = &counts[t.slot[i] * 8];
		if (c[6]) NB_THROW(ArgumentException, "Unknown base in pileup!");                                                      // Pileup.cpp:31
		if (c[7]) NB_THROW(Exception, "Could not find position " + std::to_string(snps[i].pos) + " in read!");                 // BamReader.cpp:366
		const long long depth = c[0] + c[1] + c[2] + c[3];
		if (depth < min_cov) continue;
		auto cnt = [&](char b) -> double { b = (char)toupper(b); return b == 'A' ? (double)c[0] : b == 'C' ? (double)c[1] : b == 'G' ? (double)c[2] : b == 'T' ? (double)c[3] : b == 'N' ? (double)c[4] : -1.0; };
		const double w = cnt(snps[i].ref), m = cnt(snps[i].alt);
		if (w < 0) NB_THROW(ArgumentException, std::string("Unknown wild-type base '") + snps[i].ref + "' in frequency calculation!");
		if (m < 0) NB_THROW(ArgumentException, std::string("Unknown mutant base '") + snps[i].alt + "' in frequency calculation!");
		if (w + m == 0) continue;
		++passed; passed_depth_sum += (double)depth;
		hist.inc(m / (w + m), false);
	}
	if (debug)
	{
		printf("Contamination debug output:\n%d of %zu SNPs passed quality filters\nAverage depth of passed SNPs: %s\n", passed, snps.size(), number(passed_depth_sum / passed, 2).c_str());
	}
	double off = 0.0;
	for (int i = 1; i <= 5; ++i) off += hist.binValue(i, true);
	for (int i = 14; i <= 18; ++i) off += hist.binValue(i, true);
	QCCollection output;
	Statistics::addQcValue(output, "QC:2000051", "SNV allele frequency deviation", passed < min_snps ? std::string("n/a") : number(off, 2));
	return output;
}

// somatic sub-panel numbers from the depth array selected on the reader's handle (Statistics.cpp:1574-1710)
struct SomaticDepth { long long bases_usable = 0; int hist_max = 599, hist_step = 5; std::vector<int64_t> hist; long long in_bam = 0; bool scanned = false; };
SomaticDepth somaticFromDepth(BamReader& reader, const std::vector<ngsqc_region>& regions, long long roi_bases)
{
	SomaticDepth d;
	if (!regions.empty())
	{
		std::vector<int64_t> sums(regions.size(), 0);
		reader.check(ngsqc_region_sums(reader.handle(), regions.data(), (int64_t)regions.size(), sums.data()));
		for (int64_t v : sums) d.bases_usable += v;
	}
	const double avg_depth = (double)d.bases_usable / roi_bases;
	if (avg_depth > 200) { d.hist_max += 400; d.hist_step += 5; }                                         // :1657-1670
	if (avg_depth > 500) d.hist_max += 500;
	if (avg_depth > 1000) d.hist_max += 1000;
	if (!regions.empty())
	{
		d.hist.assign((size_t)d.hist_max + 1, 0); int64_t cov = 0;
		reader.check(ngsqc_depth_stats(reader.handle(), d.hist_max, 0, d.hist.data(), &cov));
		for (auto& r : regions) d.in_bam += r.end - r.start + 1;
		d.scanned = true;
	}
	return d;
}

// state of the fused job announced by Statistics::planFused
struct FusedState
{
	bool active = false, ran = false; std::string bam; FusedPlan plan;
	SnpSites snp; std::vector<int64_t> site_counts; bool have_sites = false;
	ngsqc_read_stats rs{}; std::vector<int64_t> read_lengths, cycles; bool have_reads = false;
	SomaticDepth som; bool have_somatic = false;
	ngsqc_timings tm{}; int64_t n_blocks = 0;
};
FusedState g_fused;

struct Scan
{
	std::vector<int64_t> c = std::vector<int64_t>(NGSQC_NCOUNTERS, 0); std::vector<double> gc_reads = std::vector<double>(101, 0.0);
	int64_t operator[](int i) const { return c[(size_t)i]; }
};

bool fusedWanted(const BamReader& reader);
void runFused(BamReader& reader, const ngsqc_mapping_params& p, Scan& s);

Scan runScan(BamReader& reader, int mode, int min_mapq, const std::vector<ngsqc_region>& regions, const GcPrep* gc, const BedFile* gc_bed)
{
	Scan s; ngsqc_mapping_params p{}; p.mode = mode; p.min_mapq = min_mapq;
	p.tid_x = reader.chromosomeID(Chromosome("chrX")); p.tid_y = reader.chromosomeID(Chromosome("chrY"));
	std::vector<uint8_t> ns; for (auto& c : reader.chromosomes()) ns.push_back(c.isNonSpecial() ? 1 : 0);
	p.tid_nonspecial = ns.data();
	p.regions = regions.empty() ? nullptr : regions.data(); p.n_regions = (int64_t)regions.size();
	std::vector<ngsqc_region> chunks; std::vector<int32_t> bins;
	if (gc && gc->enabled && gc_bed)
	{
		for (long long i = 0; i < gc->dropout.count(); ++i) { int tid = reader.chromosomeID(gc->dropout[i].chr()); if (tid < 0) continue; chunks.push_back(ngsqc_region{tid, gc->dropout[i].start(), gc->dropout[i].end()}); bins.push_back(gc->bin[(size_t)i]); }
		p.gc_chunks = chunks.data(); p.gc_bin = bins.data(); p.n_gc_chunks = (int64_t)chunks.size();
	}
	const std::vector<ngsqc_handle*>& sh = reader.shards();
	if (sh.size() == 1 && fusedWanted(reader)) { runFused(reader, p, s); return s; }
	if (sh.size() == 1)
	{
		reader.check(ngsqc_scan_mapping(reader.handle(), &p, s.c.data(), s.gc_reads.data()));
		if (getenv("NGSQC_TIMING")) { ngsqc_timings tm{}; ngsqc_get_timings(reader.handle(), &tm); fprintf(stderr, "[ngsqc] mapping pass: %lld BGZF members inflated\n", (long long)tm.members_inflated); }
		return s;
	}

	// ---- one BAM sharded over several handles / GPUs (include/ngsqc.h, "sharded" section): local scans run concurrently ----
	const int n = (int)sh.size();
	auto checkShard = [&](ngsqc_handle* h, int rc) {
		if (rc == NGSQC_OK) return;
		std::string msg = ngsqc_last_error(h);
		if (rc == NGSQC_E_ARG) NB_THROW(ArgumentException, msg);
		if (rc == NGSQC_E_FORMAT) NB_THROW(FileAccessException, msg);
		NB_THROW(Exception, msg);
	};
	std::vector<ngsqc_shard_summary> sum((size_t)n); std::vector<int> rcs((size_t)n, NGSQC_OK);
	// the follow-up passes MappingQC announced (Statistics::planFused) ride every shard's decode: contamination pileup (site counts are additive over the
	// shards) and the somatic sub-panel depth (difference arrays are additive); the raw-read QC is not available on shards and keeps its own pass
	const bool fused = fusedWanted(reader);
	FusedState& F = g_fused; ngsqc_job_desc job{}; job.mapping = &p;
	std::vector<std::vector<int64_t>> site_parts; std::vector<ngsqc_region> som_regions; ngsqc_depth_params dp{}; bool with_sites = false, with_somatic = false;
	if (fused)
	{
		const FusedPlan& plan = F.plan;
		if (plan.contamination)
		{
			try
			{
				F.snp = snpSites(reader, loadKnownSnps(plan.build, plan.roi_file));
				job.sites = F.snp.sites.data(); job.n_sites = (int64_t)F.snp.sites.size();
				job.site_min_mapq = 1; job.site_min_baseq = 13; job.site_include_npp = plan.include_not_properly_paired ? 1 : 0;
				site_parts.assign((size_t)n, std::vector<int64_t>(F.snp.sites.size() * 8, 0)); with_sites = true;
			}
			catch (Exception&) { job.sites = nullptr; job.n_sites = 0; }
		}
		if (plan.somatic && plan.somatic_bed.isMergedAndSorted())
		{
			try
			{
				som_regions = toRegions(plan.somatic_bed, reader, false);
				if (!som_regions.empty()) { dp.min_mapq = plan.somatic_min_mapq; dp.regions = som_regions.data(); dp.n_regions = (int64_t)som_regions.size(); job.depth = &dp; }
				with_somatic = true;
			}
			catch (Exception&) { job.depth = nullptr; }
		}
	}
	{
		std::vector<std::thread> th;
		for (int i = 0; i < n; ++i) th.emplace_back([&, i] {
			ngsqc_job_result res{}; if (with_sites) res.site_counts = site_parts[(size_t)i].data();
			rcs[(size_t)i] = ngsqc_run_job_partial(sh[(size_t)i], &job, &res, &sum[(size_t)i]);
		});
		for (auto& t : th) t.join();
	}
	for (int i = 0; i < n; ++i) checkShard(sh[(size_t)i], rcs[(size_t)i]);
	if (fused)
	{
		if (with_sites)
		{
			F.site_counts.assign(F.snp.sites.size() * 8, 0);
			for (int i = 0; i < n; ++i) for (size_t k = 0; k < F.site_counts.size(); ++k) F.site_counts[k] += site_parts[(size_t)i][k];
			F.have_sites = true;
		}
		if (with_somatic)
		{
			if (job.depth)
			{
				for (int i = 0; i < n; ++i) checkShard(sh[(size_t)i], ngsqc_depth_select(sh[(size_t)i], 1));
				checkShard(sh[0], ngsqc_depth_reduce(sh[0], sh.data() + 1, n - 1));
				checkShard(sh[0], ngsqc_depth_finalize(sh[0]));
			}
			F.som = somaticFromDepth(reader, som_regions, F.plan.somatic_bed.baseCount());
			for (int i = 0; i < n; ++i) checkShard(sh[(size_t)i], ngsqc_depth_select(sh[(size_t)i], 0));
			F.have_somatic = true;
		}
		int64_t infl = 0, nb = 0;
		for (int i = 0; i < n; ++i) { ngsqc_timings tm{}; ngsqc_get_timings(sh[(size_t)i], &tm); infl += tm.members_inflated; nb += ngsqc_n_bgzf_blocks(sh[(size_t)i]); if (i == 0) F.tm = tm; }
		F.tm.members_inflated = infl; F.n_blocks = nb; F.ran = true;
	}
	for (int i = 0; i < n; ++i)
	{
		ngsqc_shard_fix fix{};
		if (ngsqc_plan_shard_fix(sum.data(), n, i, &fix) != NGSQC_OK) NB_THROW(FileAccessException, std::string(ngsqc_last_error(nullptr)));
		std::vector<int64_t> c((size_t)NGSQC_NCOUNTERS, 0); std::vector<double> g(101, 0.0);
		checkShard(sh[(size_t)i], ngsqc_scan_mapping_finish(sh[(size_t)i], &fix, c.data(), g.data()));
		for (int k = 0; k < NGSQC_NCOUNTERS; ++k)
		{
			const bool max_like = k == NGSQC_C_MAX_LENGTH || k == NGSQC_C_PAIRED_END || k == NGSQC_C_ROI_BASES || k == NGSQC_C_YX_VALID;
			s.c[(size_t)k] = max_like ? std::max(s.c[(size_t)k], c[(size_t)k]) : s.c[(size_t)k] + c[(size_t)k];
		}
		for (int k = 0; k < 101; ++k) s.gc_reads[(size_t)k] += g[(size_t)k];
	}
	// difference arrays are additive: summed on shard 0's GPU; shards on other GPUs are pulled with peer copies over xGMI, nothing passes
	// through host memory (a multi-process deployment all-reduces them over RCCL instead: ngs-bits_amd/dist.py)
	checkShard(sh[0], ngsqc_depth_reduce(sh[0], sh.data() + 1, n - 1));
	checkShard(sh[0], ngsqc_depth_finalize(sh[0]));
	return s;
}

// coverage-tool depth scan (ngsqc_scan_depth); with NGSQC_SHARDS the shards scan concurrently and their difference arrays are summed
void runDepthScan(BamReader& reader, const ngsqc_depth_params& p)
{
	const std::vector<ngsqc_handle*>& sh = reader.shards();
	if (sh.size() == 1)
	{
		reader.check(ngsqc_scan_depth(reader.handle(), &p));
		if (getenv("NGSQC_TIMING")) { ngsqc_timings tm{}; ngsqc_get_timings(reader.handle(), &tm); fprintf(stderr, "[ngsqc] depth pass: %lld BGZF members inflated\n", (long long)tm.members_inflated); }
		return;
	}
	const int n = (int)sh.size(); std::vector<int> rcs((size_t)n, NGSQC_OK);
	{
		std::vector<std::thread> th;
		for (int i = 0; i < n; ++i) th.emplace_back([&, i] { rcs[(size_t)i] = ngsqc_scan_depth_partial(sh[(size_t)i], &p); });
		for (auto& t : th) t.join();
	}
	auto chk = [&](ngsqc_handle* h, int rc) { if (rc == NGSQC_OK) return; std::string msg = ngsqc_last_error(h); if (rc == NGSQC_E_ARG) NB_THROW(ArgumentException, msg); if (rc == NGSQC_E_FORMAT) NB_THROW(FileAccessException, msg); NB_THROW(Exception, msg); };
	for (int i = 0; i < n; ++i) chk(sh[(size_t)i], rcs[(size_t)i]);
	chk(sh[0], ngsqc_depth_reduce(sh[0], sh.data() + 1, n - 1));   // device-side sum (peer copies between GPUs)
	chk(sh[0], ngsqc_depth_finalize(sh[0]));
}

// Statistics.cpp:576-604
void dropoutValues(const std::vector<double>& gc_roi, const std::vector<double>& gc_reads, double& at, double& gc, std::vector<double>& roi_perc, std::vector<double>& read_perc)
{
	double gc_sum = 0, roi_sum = 0; for (double v : gc_roi) gc_sum += v; for (double v : gc_reads) roi_sum += v;
	at = 0; gc = 0;
	for (int i = 0; i < 100; ++i)
	{
		double rp = 100.0 * gc_roi[(size_t)i] / gc_sum, dp = 100.0 * gc_reads[(size_t)i] / roi_sum; roi_perc.push_back(rp); read_perc.push_back(dp);
		double diff = rp - dp; if (diff > 0) { if (i <= 50) at += diff; if (i >= 50) gc += diff; }
	}
}

Histogram insertHistogram(const Scan& s) { Histogram h(0, 999, 5); for (int v = 0; v < 1000; ++v) if (s[NGSQC_C_INSERT_HIST0 + v]) h.inc(v, true, (double)s[NGSQC_C_INSERT_HIST0 + v]); return h; }

// depth histogram from the exact per-depth counts of K6 (Statistics.cpp:631-645 / 1192-1204)
Histogram depthHistogram(BamReader& reader, int hist_max, int hist_step, long long half_depth, long long& covered)
{
	std::vector<int64_t> hist((size_t)hist_max + 1, 0); int64_t cov = 0;
	reader.check(ngsqc_depth_stats(reader.handle(), hist_max, half_depth, hist.data(), &cov));
	covered = cov;
	Histogram h(0, hist_max, hist_step);
	for (int d = 0; d <= hist_max; ++d) if (hist[(size_t)d]) h.inc(d, true, (double)hist[(size_t)d]);
	return h;
}

void addYx(QCCollection& output, const Scan& s)
{
	if (s[NGSQC_C_YX_VALID]) Statistics::addQcValue(output, "QC:2000139", "chrY/chrX read ratio", number((double)s[NGSQC_C_READS_Y] / (double)s[NGSQC_C_READS_X], 4)); // Statistics.cpp:796-800
}

std::vector<double> range100() { std::vector<double> x; for (int i = 0; i < 100; ++i) x.push_back(i); return x; }

} // namespace

// ---------------------------------------------------------------- Statistics::mapping (ROI)   Statistics.cpp:343-803
QCCollection Statistics::mapping(const BedFile& bed_file, const std::string& bam_file, const std::string& ref_file, int min_mapq, bool is_cfdna)
{
	if (!bed_file.isMergedAndSorted()) NB_THROW(ArgumentException, "Merged and sorted BED file required for coverage details statistics!");
	std::unique_ptr<FastaFileIndex> fa; if (ref_file != NO_REF) fa.reset(new FastaFileIndex(ref_file));
	long long roi_bases = bed_file.baseCount();
	GcPrep gc(bed_file, fa.get());
	struct TagsGuard { int32_t old; TagsGuard() : old(ngsqc_set_cram_skip_thread(NGSQC_CRAM_SKIP_NAMES)) {} ~TagsGuard() { ngsqc_set_cram_skip_thread(old); } } tags_needed;   // (this thread's readers only; what was set before comes back)   // (the DP tag of cfDNA reads, Statistics.cpp:447-470)
	BamReader reader(bam_file, ref_file, true);
	// ROI lines on chromosomes the BAM does not know never match a read in the reference (ChromosomalIndex lookup by name)
	std::vector<ngsqc_region> regions = toRegions(bed_file, reader, false);
	Scan s = runScan(reader, NGSQC_MODE_ROI, min_mapq, regions, &gc, &bed_file);
	const double al_total = (double)s[NGSQC_C_AL_TOTAL]; const int max_length = (int)s[NGSQC_C_MAX_LENGTH]; const bool paired_end = s[NGSQC_C_PAIRED_END] != 0;
	const long long bases_usable = s[NGSQC_C_BASES_USABLE];

	double at_dropout = 0, gc_dropout = 0; std::vector<double> roi_perc, read_perc;
	dropoutValues(gc.gc_roi, s.gc_reads, at_dropout, gc_dropout, roi_perc, read_perc);

	double avg_depth = (double)bases_usable / roi_bases;
	int half_depth = (int)std::round(0.5 * avg_depth);
	int hist_max = 599, hist_step = 5;
	if (avg_depth > 200) { hist_max += 400; hist_step += 5; }
	if (avg_depth > 500) hist_max += 500;
	if (avg_depth > 1000) hist_max += 1000;
	if (is_cfdna) { hist_max = 20000; hist_step = 500; }
	long long covered_half = 0;
	// regions missing from the BAM have depth 0 everywhere: they fall into bin 0 and count for half_depth only if it is 0
	long long missing_bases = roi_bases; for (auto& r : regions) missing_bases -= (r.end - r.start + 1);
	Histogram depth_dist = depthHistogram(reader, hist_max, hist_step, half_depth, covered_half);
	if (missing_bases > 0) { depth_dist.inc(0, true, (double)missing_bases); if (0 >= half_depth) covered_half += missing_bases; }

	QCCollection output;
	addQcValue(output, "QC:2000019", "trimmed base percentage", 100.0 * (double)s[NGSQC_C_BASES_TRIMMED] / al_total / max_length);
	addQcValue(output, "QC:2000052", "clipped base percentage", 100.0 * (double)s[NGSQC_C_BASES_CLIPPED] / (double)s[NGSQC_C_BASES_MAPPED]);
	addQcValue(output, "QC:2000020", "mapped read percentage", 100.0 * s[NGSQC_C_AL_MAPPED] / al_total);
	addQcValue(output, "QC:2000021", "on-target read percentage", 100.0 * s[NGSQC_C_AL_ONTARGET] / al_total);
	addQcValue(output, "QC:2000057", "near-target read percentage", 100.0 * s[NGSQC_C_AL_NEARTARGET] / al_total);
	if (paired_end)
	{
		addQcValue(output, "QC:2000022", "properly-paired read percentage", 100.0 * s[NGSQC_C_AL_PROPER_PAIRED] / al_total);
		addQcValue(output, "QC:2000023", "insert size", (double)s[NGSQC_C_INSERT_SIZE_SUM] / (double)s[NGSQC_C_INSERT_SIZE_READ_COUNT]);
		addQcValue(output, "QC:2000150", "target region read depth (no ol)", (double)s[NGSQC_C_BASES_USABLE_NO_OVERLAP] / roi_bases);
	}
	else
	{
		addQcValue(output, "QC:2000022", "properly-paired read percentage", std::string("n/a (single end)"));
		addQcValue(output, "QC:2000023", "insert size", std::string("n/a (single end)"));
	}
	if (s[NGSQC_C_AL_DUP] == 0) addQcValue(output, "QC:2000024", "duplicate read percentage", std::string("n/a (no duplicates marked or duplicates removed during data analysis)"));
	else addQcValue(output, "QC:2000024", "duplicate read percentage", 100.0 * s[NGSQC_C_AL_DUP] / al_total);
	addQcValue(output, "QC:2000050", "bases usable (MB)", (double)bases_usable / 1000000.0);
	addQcValue(output, "QC:2000025", "target region read depth", avg_depth);

	std::vector<double> cumsum_depth(5, 0.0);
	if (is_cfdna)
	{
		double run = 0;
		for (int i = 4; i >= 0; --i) { run += (double)s[NGSQC_C_BASES_USABLE_DP0 + i] / roi_bases; cumsum_depth[(size_t)i] = run; }
		for (int i = 2; i <= 4; ++i) addQcValue(output, "QC:200007" + std::to_string(i - 1), "target region read depth " + std::to_string(i) + "-fold duplication", cumsum_depth[(size_t)i]);
		addQcValue(output, "QC:2000074", "raw target region read depth", (double)s[NGSQC_C_BASES_USABLE_RAW] / roi_bases);
	}
	std::vector<int> depths = {10, 20, 30, 50, 60, 100, 200, 500};
	std::vector<std::string> accessions = {"QC:2000026", "QC:2000027", "QC:2000028", "QC:2000029", "QC:2000099", "QC:2000030", "QC:2000031", "QC:2000032"};
	if (is_cfdna) { for (int d : {1000, 2500, 5000, 7500, 10000, 15000}) depths.push_back(d); for (const char* a : {"QC:2000065", "QC:2000066", "QC:2000067", "QC:2000068", "QC:2000069", "QC:2000070"}) accessions.push_back(a); }
	for (size_t i = 0; i < depths.size(); ++i)
	{
		double cov_bases = 0.0;
		for (int bin = depth_dist.binIndex(depths[i]); bin < depth_dist.binCount(); ++bin) cov_bases += depth_dist.binValue(bin);
		addQcValue(output, accessions[i], "target region " + std::to_string(depths[i]) + "x percentage", 100.0 * cov_bases / roi_bases);
	}
	addQcValue(output, "QC:2000058", "target region half depth percentage", 100.0 * covered_half / roi_bases);
	if (fa) { addQcValue(output, "QC:2000059", "AT dropout", at_dropout); addQcValue(output, "QC:2000060", "GC dropout", gc_dropout); }
	else { addQcValue(output, "QC:2000059", "AT dropout", std::string("n/a (no reference genome)")); addQcValue(output, "QC:2000060", "GC dropout", std::string("n/a (no reference genome)")); }

	addQcPlot(output, "QC:2000037", "depth distribution plot", depth_dist.xCoords(), {depth_dist.yCoords(true)});
	Histogram insert_dist = insertHistogram(s);
	if (paired_end) addQcPlot(output, "QC:2000038", "insert size distribution plot", insert_dist.xCoords(), {insert_dist.yCoords(true)});
	double dp_sum = 0; for (int i = 0; i < 4; ++i) dp_sum += (double)s[NGSQC_C_DP_DIST0 + i];
	if (is_cfdna && dp_sum != 0)
	{
		std::vector<double> y; for (int i = 0; i < 4; ++i) y.push_back(100.0 * (double)s[NGSQC_C_DP_DIST0 + i] / dp_sum);
		addQcPlot(output, "QC:2000075", "fragment duplication distribution plot", {1, 2, 3, 4}, {y});
		addQcPlot(output, "QC:2000076", "duplication-coverage plot", {1, 2, 3, 4}, {std::vector<double>(cumsum_depth.begin() + 1, cumsum_depth.end())});
	}
	addQcPlot(output, "QC:2000061", "GC bias plot", range100(), {roi_perc, read_perc});
	addYx(output, s);
	return output;
}

namespace {
// shared output block of the two ROI-less variants (Statistics.cpp:930-960 / 1247-1279)
void noRoiOutput(QCCollection& output, const Scan& s, bool wgs_style, double genome_size, double no_base)
{
	const double al_total = (double)s[NGSQC_C_AL_TOTAL]; const int max_length = (int)s[NGSQC_C_MAX_LENGTH]; const bool paired_end = s[NGSQC_C_PAIRED_END] != 0;
	if (!wgs_style || paired_end) Statistics::addQcValue(output, "QC:2000019", "trimmed base percentage", 100.0 * (double)s[NGSQC_C_BASES_TRIMMED] / al_total / max_length);
	else Statistics::addQcValue(output, "QC:2000019", "trimmed base percentage", std::string("n/a (single end)"));
	Statistics::addQcValue(output, "QC:2000052", "clipped base percentage", 100.0 * (double)s[NGSQC_C_BASES_CLIPPED] / (double)s[NGSQC_C_BASES_MAPPED]);
	Statistics::addQcValue(output, "QC:2000020", "mapped read percentage", 100.0 * s[NGSQC_C_AL_MAPPED] / al_total);
	Statistics::addQcValue(output, "QC:2000021", "on-target read percentage", 100.0 * s[NGSQC_C_AL_ONTARGET] / al_total);
	if (paired_end)
	{
		Statistics::addQcValue(output, "QC:2000022", "properly-paired read percentage", 100.0 * s[NGSQC_C_AL_PROPER_PAIRED] / al_total);
		Statistics::addQcValue(output, "QC:2000023", "insert size", (double)s[NGSQC_C_INSERT_SIZE_SUM] / (double)s[NGSQC_C_INSERT_SIZE_READ_COUNT]);
		Statistics::addQcValue(output, "QC:2000150", "target region read depth (no ol)", (double)s[NGSQC_C_BASES_USABLE_NO_OVERLAP] / (genome_size - no_base));
	}
	else
	{
		Statistics::addQcValue(output, "QC:2000022", "properly-paired read percentage", std::string("n/a (single end)"));
		Statistics::addQcValue(output, "QC:2000023", "insert size", std::string("n/a (single end)"));
	}
	if (s[NGSQC_C_AL_DUP] == 0) Statistics::addQcValue(output, "QC:2000024", "duplicate read percentage", std::string("n/a (duplicates not marked or removed during data analysis)"));
	else Statistics::addQcValue(output, "QC:2000024", "duplicate read percentage", 100.0 * s[NGSQC_C_AL_DUP] / al_total);
	Statistics::addQcValue(output, "QC:2000050", "bases usable (MB)", (double)s[NGSQC_C_BASES_USABLE] / 1000000.0);
	Statistics::addQcValue(output, "QC:2000025", "target region read depth", (double)s[NGSQC_C_BASES_USABLE] / (genome_size - no_base));
}
double nBases(const BamReader& reader, FastaFileIndex* fa) { double n = 0; if (fa) for (auto& c : reader.chromosomes()) if (c.isNonSpecial()) n += fa->n(c); return n; } // Statistics.cpp:920-928
}

// ---------------------------------------------------------------- Statistics::mapping(bam, ref, min_mapq)   Statistics.cpp:805-988
QCCollection Statistics::mapping(const std::string& bam_file, const std::string& ref_file, int min_mapq)
{
	BamReader reader(bam_file, ref_file, true);
	std::unique_ptr<FastaFileIndex> fa; if (ref_file != NO_REF) fa.reset(new FastaFileIndex(ref_file));
	Scan s = runScan(reader, NGSQC_MODE_NOROI, min_mapq, {}, nullptr, nullptr);
	QCCollection output;
	noRoiOutput(output, s, false, reader.genomeSize(false), nBases(reader, fa.get()));
	if (s[NGSQC_C_PAIRED_END])
	{
		Histogram insert_dist = insertHistogram(s);
		if (insert_dist.binSum() > 0) addQcPlot(output, "QC:2000038", "insert size distribution plot", insert_dist.xCoords(), {insert_dist.yCoords(true)});
		else fprintf(stderr, "Skipping insert size histogram - no read pairs found!\n");
	}
	addYx(output, s);
	return output;
}

// ---------------------------------------------------------------- Statistics::mapping_wgs   Statistics.cpp:990-1359
QCCollection Statistics::mapping_wgs(const std::string& bam_file, const std::string& bedpath, int min_mapq, const std::string& ref_file)
{
	BamReader reader(bam_file, ref_file, true);
	std::unique_ptr<FastaFileIndex> fa; if (ref_file != NO_REF) fa.reset(new FastaFileIndex(ref_file));
	bool roi_available = false; BedFile roi;
	if (!bedpath.empty())
	{
		roi_available = true; roi.load(bedpath, false);
		if (!roi.isMergedAndSorted()) { roi.sort(); roi.merge(); }
	}
	GcPrep gc(roi, fa.get());
	std::vector<ngsqc_region> regions;
	if (roi_available) { reader.requireIndex(); regions = toRegions(roi, reader, true); }
	Scan s = runScan(reader, NGSQC_MODE_WGS, min_mapq, regions, &gc, &roi);

	QCCollection output;
	noRoiOutput(output, s, true, reader.genomeSize(false), nBases(reader, fa.get()));
	Histogram insert_dist = insertHistogram(s);
	if (roi_available)
	{
		const double roi_bases = (double)roi.baseCount();
		double avg_depth = (double)s[NGSQC_C_BASES_USABLE_ROI] / roi_bases;
		int half_depth = (int)std::round(0.5 * avg_depth);
		long long covered_half = 0;
		Histogram depth_dist = depthHistogram(reader, 599, 5, half_depth, covered_half);
		double at_dropout = 0, gc_dropout = 0; std::vector<double> roi_perc, read_perc;
		dropoutValues(gc.gc_roi, s.gc_reads, at_dropout, gc_dropout, roi_perc, read_perc);
		const int depth_values[8] = {10, 20, 30, 50, 60, 100, 200, 500};
		const char* accessions[8] = {"QC:2000026", "QC:2000027", "QC:2000028", "QC:2000029", "QC:2000099", "QC:2000030", "QC:2000031", "QC:2000032"};
		for (int i = 0; i < 8; ++i)
		{
			double cov_bases = 0.0;
			for (int bin = depth_dist.binIndex(depth_values[i]); bin < depth_dist.binCount(); ++bin) cov_bases += depth_dist.binValue(bin);
			addQcValue(output, accessions[i], "target region " + std::to_string(depth_values[i]) + "x percentage", 100.0 * cov_bases / roi_bases);
		}
		addQcValue(output, "QC:2000058", "target region half depth percentage", 100.0 * covered_half / roi_bases);
		if (fa) { addQcValue(output, "QC:2000059", "AT dropout", at_dropout); addQcValue(output, "QC:2000060", "GC dropout", gc_dropout); }
		else { addQcValue(output, "QC:2000059", "AT dropout", std::string("n/a (no reference genome)")); addQcValue(output, "QC:2000060", "GC dropout", std::string("n/a (no reference genome)")); }
		addQcPlot(output, "QC:2000037", "depth distribution plot", depth_dist.xCoords(), {depth_dist.yCoords(true)});
		if (s[NGSQC_C_PAIRED_END])
		{
			if (insert_dist.binSum() > 0) addQcPlot(output, "QC:2000038", "insert size distribution plot", insert_dist.xCoords(), {insert_dist.yCoords(true)});
			else fprintf(stderr, "Skipping insert size histogram - no read pairs found!\n");
		}
		addQcPlot(output, "QC:2000061", "GC bias plot", range100(), {roi_perc, read_perc});
	}
	else if (s[NGSQC_C_PAIRED_END])
	{
		if (insert_dist.binSum() > 0) addQcPlot(output, "QC:2000038", "insert size distribution plot", insert_dist.xCoords(), {insert_dist.yCoords(true)});
		else fprintf(stderr, "Skipping insert size histogram - no read pairs found!\n");
	}
	addYx(output, s);
	return output;
}

// ---------------------------------------------------------------- coverage tools
namespace {
// merged + sorted union of the BED lines (the scan's depth array), restricted to chromosomes known to the BAM
std::vector<ngsqc_region> unionRegions(const BedFile& bed, const BamReader& reader, bool back_to_back)
{
	BedFile u; u.add(bed); u.clearAnnotations(); u.merge(back_to_back);
	return toRegions(u, reader, true);
}
}

// Statistics.cpp:1574-1710. The reference keeps a QMap<position, depth> per chromosome; the read filter (no secondary /
// supplementary / unmapped / duplicate, MAPQ >= min_mapq) and the whole-reference-span increment are those of the coverage
// tools, so the GPU side is ngsqc_scan_depth on the (merged) sub-panel: bases_usable = sum of the per-base depths, the
// histogram comes from the exact per-depth counts of K6.
QCCollection Statistics::somaticCustomDepth(const BedFile& bed_file, const std::string& bam_file, const std::string& ref_file, int min_mapq)
{
	if (!bed_file.isMergedAndSorted()) NB_THROW(ArgumentException, "Merged and sorted BED file required for depth details statistics!");   // :1577-1580
	long long roi_bases = bed_file.baseCount();
	SomaticDepth d;
	if (g_fused.ran && g_fused.have_somatic && g_fused.bam == bam_file && g_fused.plan.somatic_min_mapq == min_mapq) d = g_fused.som;   // the fused job scanned the sub-panel beside the mapping scan
	else
	{
		BamReader reader(bam_file, ref_file, true);
		std::vector<ngsqc_region> regions = toRegions(bed_file, reader, false);
		if (!regions.empty())
		{
			ngsqc_depth_params p{}; p.min_mapq = min_mapq; p.min_baseq = 0; p.skip_mismapped = 0; p.regions = regions.data(); p.n_regions = (int64_t)regions.size();
			runDepthScan(reader, p);
		}
		d = somaticFromDepth(reader, regions, roi_bases);
	}
	double avg_depth = (double)d.bases_usable / roi_bases;
	Histogram depth_dist(0, d.hist_max, d.hist_step);
	if (d.scanned) for (int k = 0; k <= d.hist_max; ++k) if (d.hist[(size_t)k]) depth_dist.inc(k, true, (double)d.hist[(size_t)k]);
	if (roi_bases > d.in_bam) depth_dist.inc(0, true, (double)(roi_bases - d.in_bam));   // positions on chromosomes the BAM does not know stay at depth 0
	QCCollection output;
	addQcValue(output, "QC:2000097", "somatic custom target region read depth", avg_depth);
	const int depths[8] = {10, 20, 30, 50, 60, 100, 200, 500};
	const char* accessions[8] = {"QC:2000090", "QC:2000091", "QC:2000092", "QC:2000093", "QC:2000098", "QC:2000094", "QC:2000095", "QC:2000096"};
	for (int i = 0; i < 8; ++i)
	{
		double cov_bases = 0.0;
		for (int bin = depth_dist.binIndex(depths[i]); bin < depth_dist.binCount(); ++bin) cov_bases += depth_dist.binValue(bin);
		addQcValue(output, accessions[i], "somatic custom target " + std::to_string(depths[i]) + "x percentage", 100.0 * cov_bases / roi_bases);
	}
	return output;
}

// ---------------------------------------------------------------- StatisticsReads (StatisticsReads.cpp:83-442)
void StatisticsReads::update(BamReader& reader)
{
	reader.check(ngsqc_scan_reads(reader.handle(), single_end_ ? 1 : 0, &st_));
	if (getenv("NGSQC_TIMING")) { ngsqc_timings tm{}; ngsqc_get_timings(reader.handle(), &tm); fprintf(stderr, "[ngsqc] read QC pass: %lld BGZF members inflated\n", (long long)tm.members_inflated); }
	if (st_.n_unknown_base) NB_THROW(ProgrammingException, "Unknown base in StatisticsReads::update!");                                    // :131
	if (st_.n_quality_out_of_range) NB_THROW(ArgumentException, "Base quality > 100. This should not happen!");                            // :141
	read_lengths_.assign((size_t)st_.max_cycles + 1, 0);
	reader.check(ngsqc_read_length_hist(reader.handle(), read_lengths_.data(), (int64_t)read_lengths_.size()));
	const int64_t n_cyc = std::min<int64_t>(st_.max_cycles, 320);
	cycles_.assign((size_t)n_cyc * 7, 0);
	if (n_cyc) reader.check(ngsqc_read_cycle_stats(reader.handle(), cycles_.data(), n_cyc));
	have_ = true;
}

bool StatisticsReads::takeFused(const std::string& bam_file)
{
	if (!(g_fused.ran && g_fused.have_reads && g_fused.bam == bam_file && g_fused.plan.single_end == single_end_)) return false;
	st_ = g_fused.rs; read_lengths_ = g_fused.read_lengths; cycles_ = g_fused.cycles;
	if (st_.n_unknown_base) NB_THROW(ProgrammingException, "Unknown base in StatisticsReads::update!");                                    // :131
	if (st_.n_quality_out_of_range) NB_THROW(ArgumentException, "Base quality > 100. This should not happen!");                            // :141
	have_ = true;
	return true;
}

QCCollection StatisticsReads::getResult()
{
	QCCollection output;
	auto put = [&](const std::string& name, const std::string& value, const std::string& desc, const std::string& acc) { QCValue v; v.name = name; v.accession = acc; v.description = desc; v.type = QCValueType::STRING; v.s = value; output.insert(v); };
	auto putd = [&](const std::string& name, double value, const std::string& desc, const std::string& acc) { QCValue v; v.name = name; v.accession = acc; v.description = desc; v.type = QCValueType::DOUBLE; v.d = value; output.insert(v); };
	auto putp = [&](const std::string& name, const std::vector<double>& x, const std::vector<std::vector<double>>& lines, const std::string& desc, const std::string& acc) { QCValue v; v.name = name; v.accession = acc; v.description = desc; v.type = QCValueType::IMAGE; v.s = plotPng(x, lines); output.insert(v); };
	const long long c_forward = st_.c_forward, c_reverse = st_.c_reverse, bases_sequenced = st_.bases_sequenced;
	const long long total_reads = c_forward + c_reverse;
	const long long c_base_n = st_.bases[4], c_base_gc = st_.bases[1] + st_.bases[2];
	const long long bases_total = st_.bases[0] + st_.bases[1] + st_.bases[2] + st_.bases[3] + st_.bases[4];       // sum of pileup.depth(false, true)
	long long c_base_q20 = 0, c_base_q30 = 0, c_read_q20 = 0;
	for (int q = 20; q < 100; ++q) c_base_q20 += st_.base_qualities[q];
	for (int q = 30; q < 100; ++q) c_base_q30 += st_.base_qualities[q];
	for (int b = 20; b < 60; ++b) c_read_q20 += st_.qscore_dist_r1[b] + st_.qscore_dist_r2[b];                  // mean >= 20  <=>  bin >= 20 of Histogram(0,60,1)
	std::vector<int> tmp; for (size_t l = 0; l < read_lengths_.size(); ++l) if (read_lengths_[l]) tmp.push_back((int)l);          // read_lengths_.keys()
	if (tmp.empty()) tmp.push_back(0);
	const int longest_read = tmp.back();
	const bool is_longread = single_end_ && longest_read >= 10000;
	put("read count", std::to_string(total_reads), "Total number of reads (forward and reverse reads of paired-end sequencing count as two reads).", "QC:2000005");
	std::string lengths;
	if (tmp.size() < 4) { lengths = std::to_string(tmp[0]); for (size_t i = 1; i < tmp.size(); ++i) lengths += ", " + std::to_string(tmp[i]); }
	else lengths = std::to_string(tmp[0]) + "-" + std::to_string(longest_read);
	put("read length", lengths, "Raw read length of a single read before trimming. Comma-separated list of lenghs or length range, if reads have different lengths.", "QC:2000006");
	putd("bases sequenced (MB)", (double)bases_sequenced / 1000000.0, "Bases sequenced in total (in megabases).", "QC:2000049");
	putd("Q20 read percentage", 100.0 * c_read_q20 / total_reads, "The percentage of reads with a mean base quality score greater than Q20.", "QC:2000007");
	putd("Q20 base percentage", 100.0 * c_base_q20 / bases_total, "The percentage of bases with a minimum quality score of Q20.", "QC:2000148");
	putd("Q30 base percentage", 100.0 * c_base_q30 / bases_total, "The percentage of bases with a minimum quality score of Q30.", "QC:2000008");
	putd("no base call percentage", 100.0 * c_base_n / bases_total, "The percentage of bases without base call (N).", "QC:2000009");
	putd("gc content percentage", 100.0 * c_base_gc / (bases_total - c_base_n), "The percentage of bases that are called to be G or C.", "QC:2000010");
	if (single_end_)   // N50 (:190-208)
	{
		long long bases = 0; int n50 = 0;
		for (size_t k = tmp.size(); k-- > 0;) { bases += (long long)tmp[k] * read_lengths_[(size_t)tmp[k]]; if (bases > bases_sequenced / 2) { n50 = tmp[k]; break; } }
		put("N50 read length (bp)", std::to_string(n50), "Minimum read length to reach 50% of sequenced bases.", "QC:2000131");
	}
	int n95 = -1;      // :211-237
	if (is_longread)
	{
		long long bases = 0;
		for (int l : tmp) { bases += (long long)l * read_lengths_[(size_t)l]; if (bases > 0.95 * bases_sequenced) { n95 = l; break; } }
		if (longest_read <= 100000) n95 = (int)(std::ceil(n95 / 1000.0) * 1000); else n95 = (int)(std::ceil(n95 / 10000.0) * 10000);
	}
	// per-cycle plots (the GPU pass keeps per-cycle statistics for the first 320 cycles)
	int cycles = longest_read; if (is_longread) cycles = std::min(n95, cycles);
	cycles = std::min<int>(cycles, (int)(cycles_.size() / 7));
	std::vector<double> la, lc, lg, lt, ln, lgc, lx, q1, q2;
	for (int i = 0; i < cycles; ++i)
	{
		const int64_t* c = &cycles_[(size_t)i * 7];
		const double depth_no_n = (double)(c[0] + c[1] + c[2] + c[3]);
		la.push_back(100.0 * c[0] / depth_no_n); lc.push_back(100.0 * c[1] / depth_no_n); lg.push_back(100.0 * c[2] / depth_no_n); lt.push_back(100.0 * c[3] / depth_no_n);
		ln.push_back(100.0 * c[4] / (depth_no_n + c[4])); lgc.push_back(lg.back() + lc.back()); lx.push_back(i + 1);
		long long depth = c[0] + c[1] + c[2] + c[3] + c[4]; if (c_reverse > 0) depth /= 2;
		q1.push_back((double)c[5] / depth); q2.push_back((double)c[6] / depth);
	}
	putp("base distribution plot", lx, {la, lc, lg, lt, ln, lgc}, "Base distribution plot per cycle.", "QC:2000011");
	if (c_reverse > 0) putp("Q score plot", lx, {q1, q2}, "Mean Q score per cycle for forward/reverse reads.", "QC:2000012");
	else putp("Q score plot", lx, {q1}, "Mean Q score per cycle for forward/reverse reads.", "QC:2000012");
	{
		std::vector<double> x, y1, y2; long long s1 = 0, s2 = 0;
		for (int b = 0; b < 60; ++b) { s1 += st_.qscore_dist_r1[b]; s2 += st_.qscore_dist_r2[b]; }
		for (int b = 0; b < 60; ++b) { x.push_back(b + 0.5); y1.push_back(100.0 * st_.qscore_dist_r1[b] / s1); y2.push_back(100.0 * st_.qscore_dist_r2[b] / s2); }
		if (c_reverse > 0) putp("read Q score distribution", x, {y1, y2}, "Distrubition of the mean forward/reverse Q score for each read.", "QC:2000138");
		else putp("read Q score distribution", x, {y1}, "Distrubition of the mean forward/reverse Q score for each read.", "QC:2000138");
	}
	if (single_end_)   // :342-438
	{
		const int hist_min = std::max(0, tmp.front() - 20), hist_max = (is_longread ? n95 : longest_read) + 20;
		const int step = std::max(1, (hist_max - hist_min) / 60);
		Histogram read_length_hist(hist_min, hist_max, step);
		for (int l : tmp) read_length_hist.inc(l, true, (double)read_lengths_[(size_t)l]);
		putp("Read length histogram", read_length_hist.xCoords(), {read_length_hist.yCoords(true)}, "Histogram of read lengths", "QC:2000132");
		std::vector<double> xs, values; long long max_count = 0, bases_checked = 0; int mode_base_q_score = 0, median_base_q_score = -1;
		for (int i = 0; i <= 60; ++i)
		{
			const long long base_count = st_.base_qualities[i];
			xs.push_back(i); values.push_back((100.0 * base_count) / bases_sequenced);
			if (base_count >= max_count) { max_count = base_count; if (i < 50) mode_base_q_score = i; }
			bases_checked += base_count;
			if (median_base_q_score == -1 && bases_checked * 2 >= bases_sequenced) median_base_q_score = i;
		}
		putp("base Q score histogram", xs, {values}, "Histogram of base Q scores.", "QC:2000143");
		put("median base Q score", std::to_string(median_base_q_score), "Median Q score of all bases of the sample.", "QC:2000144");
		put("mode base Q score", std::to_string(mode_base_q_score), "Most frequent Q score of all bases of the sample.", "QC:2000145");
		max_count = 0; int mode_read_q_score = 0, median_read_q_score = -1; long long reads_checked = 0;
		for (int i = 0; i < 100; ++i)
		{
			const long long read_count = st_.read_qualities[i];
			if (read_count >= max_count) { max_count = read_count; mode_read_q_score = i; }
			reads_checked += read_count;
			if (median_read_q_score == -1 && reads_checked * 2 >= c_forward) median_read_q_score = i;
		}
		put("median read Q score", std::to_string(median_read_q_score), "Median Q score of all reads of the sample.", "QC:2000146");
		put("mode read Q score", std::to_string(mode_read_q_score), "Most frequent Q score of all reads of the sample.", "QC:2000147");
	}
	return output;
}

// Statistics.cpp:2333-2386 + NGSHelper::getKnownVariants (NGSHelper.cpp:22-94) + BamReader::getPileup (BamReader.cpp:809-885).
// The reference runs one indexed pileup query per known SNP; here all sites go to the GPU in one table (ngsqc_site_pileup, or the
// site-pileup consumer of the fused job).
namespace {
bool fusedWanted(const BamReader& reader) { return g_fused.active && !g_fused.ran && g_fused.bam == reader.fileName(); }

// ONE pass over the BAM for the mapping scan and every announced follow-up pass (ngsqc_run_job)
void runFused(BamReader& reader, const ngsqc_mapping_params& p, Scan& s)
{
	FusedState& F = g_fused; const FusedPlan& plan = F.plan;
	ngsqc_job_desc job{}; ngsqc_job_result res{};
	job.mapping = &p; res.counters = s.c.data(); res.gc_reads = s.gc_reads.data();
	// A follow-up pass whose inputs are broken (a known SNP on a chromosome the BAM does not have, an unmerged sub-panel ...) is left out of the job:
	// its own call throws later - behind the read QC output, where the reference's pass order puts the error (src/MappingQC/main.cpp:80-165).
	bool with_sites = false, with_somatic = false;
	if (plan.contamination)
	{
		try
		{
			F.snp = snpSites(reader, loadKnownSnps(plan.build, plan.roi_file));
			F.site_counts.assign(F.snp.sites.size() * 8, 0);
			job.sites = F.snp.sites.data(); job.n_sites = (int64_t)F.snp.sites.size();
			job.site_min_mapq = 1; job.site_min_baseq = 13; job.site_include_npp = plan.include_not_properly_paired ? 1 : 0;
			res.site_counts = F.site_counts.data(); with_sites = true;
		}
		catch (Exception&) { job.sites = nullptr; job.n_sites = 0; res.site_counts = nullptr; }
	}
	if (plan.read_qc) { job.read_qc = 1; job.read_qc_single_end = plan.single_end ? 1 : 0; res.read_stats = &F.rs; }
	std::vector<ngsqc_region> som_regions; ngsqc_depth_params dp{};
	if (plan.somatic && plan.somatic_bed.isMergedAndSorted())   // (unmerged: Statistics::somaticCustomDepth throws, Statistics.cpp:1577-1580)
	{
		try
		{
			som_regions = toRegions(plan.somatic_bed, reader, false);
			if (!som_regions.empty()) { dp.min_mapq = plan.somatic_min_mapq; dp.regions = som_regions.data(); dp.n_regions = (int64_t)som_regions.size(); job.depth = &dp; }
			with_somatic = true;
		}
		catch (Exception&) { job.depth = nullptr; }
	}
	stamp("fused job: start");
	reader.check(ngsqc_run_job(reader.handle(), &job, &res));
	stamp("fused job: done");
	ngsqc_get_timings(reader.handle(), &F.tm); F.n_blocks = ngsqc_n_bgzf_blocks(reader.handle());
	F.have_sites = with_sites;
	if (plan.read_qc)
	{
		F.read_lengths.assign((size_t)F.rs.max_cycles + 1, 0);
		reader.check(ngsqc_read_length_hist(reader.handle(), F.read_lengths.data(), (int64_t)F.read_lengths.size()));
		const int64_t n_cyc = std::min<int64_t>(F.rs.max_cycles, 320);
		F.cycles.assign((size_t)n_cyc * 7, 0);
		if (n_cyc) reader.check(ngsqc_read_cycle_stats(reader.handle(), F.cycles.data(), n_cyc));
		F.have_reads = true;
	}
	if (with_somatic)
	{
		if (job.depth) reader.check(ngsqc_depth_select(reader.handle(), 1));
		F.som = somaticFromDepth(reader, som_regions, plan.somatic_bed.baseCount());
		reader.check(ngsqc_depth_select(reader.handle(), 0));
		F.have_somatic = true;
	}
	F.ran = true;
}
} // namespace

void Statistics::planFused(const std::string& bam_file, const FusedPlan& plan)
{
	g_fused = FusedState();
	const char* e = getenv("NGSQC_FUSED");
	if (e && atoi(e) == 0) return;
	g_fused.active = true; g_fused.bam = bam_file; g_fused.plan = plan;
}
void Statistics::clearFused()
{
	if (g_fused.ran && getenv("NGSQC_TIMING"))
		fprintf(stderr, "[ngsqc] fused job: %.2f ms wall, K1 %.2f ms, %lld tiles, %lld of %lld BGZF members inflated, %lld records\n", g_fused.tm.job_wall_ms, g_fused.tm.inflate_ms,
		        (long long)g_fused.tm.n_tiles, (long long)g_fused.tm.members_inflated, (long long)g_fused.n_blocks, (long long)g_fused.tm.n_records);
	g_fused = FusedState();
}

QCCollection Statistics::contamination(const std::string& build, const std::string& bam, const std::string& ref_file, const std::string& roi_file, bool debug, int min_cov, int min_snps, bool include_not_properly_paired)
{
	if (g_fused.ran && g_fused.have_sites && g_fused.bam == bam && g_fused.plan.build == build && g_fused.plan.roi_file == roi_file && g_fused.plan.include_not_properly_paired == include_not_properly_paired)
		return contaminationFromCounts(g_fused.snp, g_fused.site_counts, debug, min_cov, min_snps);
	BamReader reader(bam, ref_file);
	SnpSites t = snpSites(reader, loadKnownSnps(build, roi_file));
	std::vector<int64_t> counts(t.sites.size() * 8, 0);
	if (!t.sites.empty()) reader.check(ngsqc_site_pileup(reader.handle(), t.sites.data(), (int64_t)t.sites.size(), 1, 13, include_not_properly_paired ? 1 : 0, counts.data()));
	if (getenv("NGSQC_TIMING")) { ngsqc_timings tm{}; ngsqc_get_timings(reader.handle(), &tm); fprintf(stderr, "[ngsqc] contamination pass: %lld BGZF members inflated\n", (long long)tm.members_inflated); }
	return contaminationFromCounts(t, counts, debug, min_cov, min_snps);
}

// ---------------------------------------------------------------- SampleGender (Statistics.cpp:2811-2902)
GenderEstimate Statistics::genderXY(const std::string& bam_file, double max_female, double min_male, const std::string& ref_file)
{
	BamReader reader(bam_file, ref_file);
	// Statistics::yxRatio (:2659-2691): two indexed passes over chrY and chrX in the reference, two counters of one scan here
	double count_x = 0.0, count_y = 0.0, ratio_yx = std::numeric_limits<double>::quiet_NaN();
	if (reader.chromosomeID(Chromosome("chrX")) >= 0 && reader.chromosomeID(Chromosome("chrY")) >= 0)
	{
		reader.requireIndex();   // setRegion (BamReader.cpp:740-746)
		Scan s = runScan(reader, NGSQC_MODE_NOROI, 1, {}, nullptr, nullptr);
		count_x = (double)s[NGSQC_C_READS_X]; count_y = (double)s[NGSQC_C_READS_Y];
		if (count_x != 0) ratio_yx = count_y / count_x;
	}
	GenderEstimate output;
	output.add_info.push_back({"reads_chry", number(count_y, 0)});
	output.add_info.push_back({"reads_chrx", number(count_x, 0)});
	output.add_info.push_back({"ratio_chry_chrx", std::isnan(ratio_yx) ? std::string("nan") : number(ratio_yx, 4)});
	if (ratio_yx <= max_female) output.gender = "female";
	else if (ratio_yx >= min_male) output.gender = "male";
	else output.gender = "unknown (ratio in gray area)";
	return output;
}

GenderEstimate Statistics::genderHetX(const std::string& build, const std::string& bam_file, double max_male, double min_female, const std::string& ref_file, bool include_not_properly_paired)
{
	BamReader reader(bam_file, ref_file);
	// common SNPs on chrX outside the pseudo-autosomal regions (NGSHelper::pseudoAutosomalRegion, NGSHelper.cpp:415-434)
	Chromosome chrx("chrX");
	const int chrx_end_pos = reader.chromosomeSize(chrx);
	const bool hg38 = build == "hg38";
	const int par[2][2] = {{hg38 ? 10001 : 60001, hg38 ? 2781479 : 2699520}, {hg38 ? 155701383 : 154931044, hg38 ? 156030895 : 155260560}};
	BedFile roi_chrx;   // chrX:[1, end] minus the two PAR lines (BedFile::subtract)
	int s0 = 1;
	for (int k = 0; k < 2; ++k) { if (par[k][0] > s0) roi_chrx.append(BedLine(chrx, s0, std::min(par[k][0] - 1, chrx_end_pos))); s0 = std::max(s0, par[k][1] + 1); }
	if (s0 <= chrx_end_pos) roi_chrx.append(BedLine(chrx, s0, chrx_end_pos));
	SnpSites t = snpSites(reader, loadKnownSnps(build, "", &roi_chrx));
	std::vector<int64_t> counts(t.sites.size() * 8, 0);
	if (!t.sites.empty()) reader.check(ngsqc_site_pileup(reader.handle(), t.sites.data(), (int64_t)t.sites.size(), 20, 20, include_not_properly_paired ? 1 : 0, counts.data()));
	int c_all = 0, c_het = 0;
	for (size_t i = 0; i < t.snps.size(); ++i)
	{
		const int64_t* c = &counts[t.slot[i] * 8];
		if (c[6]) NB_THROW(ArgumentException, "Unknown base in pileup!");
		if (c[7]) NB_THROW(Exception, "Could not find position " + std::to_string(t.snps[i].pos) + " in read!");
		const long long depth = c[0] + c[1] + c[2] + c[3];
		if (depth < 20) continue;
		auto cnt = [&](char b) -> double { b = (char)toupper(b); return b == 'A' ? (double)c[0] : b == 'C' ? (double)c[1] : b == 'G' ? (double)c[2] : b == 'T' ? (double)c[3] : b == 'N' ? (double)c[4] : -1.0; };
		const double w = cnt(t.snps[i].ref), m = cnt(t.snps[i].alt);
		if (w < 0) NB_THROW(ArgumentException, std::string("Unknown wild-type base '") + t.snps[i].ref + "' in frequency calculation!");
		if (m < 0) NB_THROW(ArgumentException, std::string("Unknown mutant base '") + t.snps[i].alt + "' in frequency calculation!");
		if (w + m == 0) continue;   // frequency() is NaN
		const double af = m / (w + m);
		++c_all;
		if (af > 0.1 && af < 0.9) ++c_het;
	}
	const double het_frac = (double)c_het / c_all;
	GenderEstimate output;
	output.add_info.push_back({"snps_usable", std::to_string(c_all) + " of " + std::to_string(t.snps.size())});
	output.add_info.push_back({"hom_count", std::to_string(c_all - c_het)});
	output.add_info.push_back({"het_count", std::to_string(c_het)});
	output.add_info.push_back({"het_fraction", std::isnan(het_frac) ? std::string("nan") : number(het_frac, 4)});
	if (c_all < 20) output.gender = "unknown (too few SNPs)";
	else if (het_frac <= max_male) output.gender = "male";
	else if (het_frac >= min_female) output.gender = "female";
	else output.gender = "unknown (fraction in gray area)";
	return output;
}

GenderEstimate Statistics::genderSRY(const std::string& build, const std::string& bam_file, double min_cov, const std::string& ref_file)
{
	const bool hg38 = build == "hg38";
	BedFile roi; roi.append(BedLine(Chromosome("chrY"), hg38 ? 2786989 : 2655031, hg38 ? 2787603 : 2655641));
	Statistics::avgCoverage(roi, bam_file, 1, 1, 2, ref_file);
	const double cov = atof(roi[0].annotations()[0].c_str());
	GenderEstimate output;
	output.add_info.push_back({"coverage_sry", number(cov, 2)});
	output.gender = cov >= min_cov ? "male" : "female";
	return output;
}

// Lines that lie far apart in the file (a handful of genes spread over the genome: BedCoverage -random_access, WorkerAverageCoverage.cpp:100-173 queries the index per
// line) are served by one index-driven handle per CLUSTER of lines instead of one handle over the range from the first line to the last: the per-line ranges of the
// BAI (ngsqc_bai_ranges) are sorted and merged while they overlap or lie closer than 8 MB of compressed bytes. Returns false when that does not pay (no index, one
// cluster, too many lines or clusters): the caller takes the single-handle path.
static bool avgCoverageClustered(BedFile& bed_file, const std::string& bam_file, int min_mapq, int decimals, const std::string& ref_file, bool skip_mismapped, int threads)
{
	const char* es = getenv("NGSQC_INDEX_SELECT");
	if ((es && atoi(es) == 0) || bed_file.count() < 2 || bed_file.count() > 4096 || getenv("NGSQC_SHARDS")) return false;
	if (!hasBamIndex(bam_file)) return false;
	if (bam_file.size() > 5 && bam_file.compare(bam_file.size() - 5, 5, ".cram") == 0) return false;   // (a CRAM: the library picks the slices of the regions itself)
	std::vector<ngsqc_region> lines; int n_ref = 0;
	{
		BamReader head(bam_file, ref_file, BamReader::Head{1});   // (chromosome numbering of this BAM: the header members only)
		lines = toRegions(bed_file, head, true); n_ref = ngsqc_n_ref(head.handle());
	}
	const size_t n = lines.size();
	std::vector<uint64_t> vb(n, 0), ve(n, 0);
	if (ngsqc_bai_ranges(bam_file.c_str(), lines.data(), (int64_t)n, n_ref, vb.data(), ve.data()) != NGSQC_OK) return false;
	std::vector<size_t> order; for (size_t i = 0; i < n; ++i) if (ve[i]) order.push_back(i);
	std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return vb[a] != vb[b] ? vb[a] < vb[b] : a < b; });
	std::vector<std::vector<size_t>> clusters; uint64_t cur_end = 0;
	uint64_t GAP = 8ull << 20; if (const char* eg = getenv("NGSQC_INDEX_CLUSTER_GAP_KB")) GAP = (uint64_t)std::max(1, atoi(eg)) << 10;   // (tests: small files)
	for (size_t i : order)
	{
		if (clusters.empty() || (vb[i] >> 16) > (cur_end >> 16) + GAP) { clusters.emplace_back(); cur_end = 0; }
		clusters.back().push_back(i); cur_end = std::max(cur_end, ve[i]);
	}
	if (clusters.size() < 2 || clusters.size() > 64) return false;
	std::vector<int64_t> sums(n, 0);
	// -threads (Statistics.cpp:2778-2797: the reference runs its random-access chunks in a QThreadPool(threads), one BamReader each): the clusters are independent
	// index-driven handles with streams of their own, up to `threads` of them at work at once; every cluster writes its own lines' sums - the result does not
	// depend on the number of threads, as the reference's tests ask
	auto one = [&](const std::vector<size_t>& cl) {
		BedFile sub; for (size_t i : cl) sub.append(bed_file[(long long)i]);
		BamReader reader(bam_file, ref_file, true, sub);
		reader.requireIndex();
		std::vector<ngsqc_region> regions = unionRegions(sub, reader, true);
		ngsqc_depth_params p{}; p.min_mapq = min_mapq; p.min_baseq = 0; p.skip_mismapped = skip_mismapped ? 1 : 0; p.regions = regions.data(); p.n_regions = (int64_t)regions.size();
		runDepthScan(reader, p);
		std::vector<ngsqc_region> sl = toRegions(sub, reader, true);
		std::vector<int64_t> ss(sl.size(), 0);
		reader.check(ngsqc_region_sums(reader.handle(), sl.data(), (int64_t)sl.size(), ss.data()));
		for (size_t k = 0; k < cl.size(); ++k) sums[cl[k]] = ss[k];
	};
	const int workers = (int)std::min<size_t>((size_t)std::max(1, threads), clusters.size());
	if (workers <= 1) for (const std::vector<size_t>& cl : clusters) one(cl);
	else
	{
		std::atomic<size_t> next(0); std::mutex mu; std::exception_ptr first;
		std::vector<std::thread> pool;
		for (int t = 0; t < workers; ++t) pool.emplace_back([&] {
			for (;;)
			{
				const size_t i = next.fetch_add(1);
				if (i >= clusters.size()) return;
				{ std::lock_guard<std::mutex> g(mu); if (first) return; }
				try { one(clusters[i]); } catch (...) { std::lock_guard<std::mutex> g(mu); if (!first) first = std::current_exception(); return; }
			}
		});
		for (auto& t : pool) t.join();
		if (first) std::rethrow_exception(first);
	}
	if (getenv("NGSQC_TIMING")) fprintf(stderr, "[ngsqc] index-driven open: %zu clusters of lines\n", clusters.size());
	for (long long i = 0; i < bed_file.count(); ++i) bed_file[i].annotations().push_back(number((double)sums[(size_t)i] / bed_file[i].length(), decimals));
	return true;
}

void Statistics::avgCoverage(BedFile& bed_file, const std::string& bam_file, int min_mapq, int threads, int decimals, const std::string& ref_file, bool random_access, bool skip_mismapped, bool /*debug*/)
{
	if (!random_access && !bed_file.isSorted()) NB_THROW(ArgumentException, "Input BED file has to be sorted for sweep algorithm!");
	if (bed_file.count() == 0) return;
	if (avgCoverageClustered(bed_file, bam_file, min_mapq, decimals, ref_file, skip_mismapped, threads)) return;
	BamReader reader(bam_file, ref_file, true, bed_file);   // (only the BGZF blocks the index names for the lines)
	reader.requireIndex();
	std::vector<ngsqc_region> regions = unionRegions(bed_file, reader, true);
	ngsqc_depth_params p{}; p.min_mapq = min_mapq; p.min_baseq = 0; p.skip_mismapped = skip_mismapped ? 1 : 0; p.regions = regions.data(); p.n_regions = (int64_t)regions.size();
	runDepthScan(reader, p);
	std::vector<ngsqc_region> lines = toRegions(bed_file, reader, true);
	std::vector<int64_t> sums(lines.size(), 0);
	reader.check(ngsqc_region_sums(reader.handle(), lines.data(), (int64_t)lines.size(), sums.data()));
	// sum of read/line overlaps == sum of per-base depth over the line (WorkerAverageCoverage.cpp:47-55,135-155)
	for (long long i = 0; i < bed_file.count(); ++i) bed_file[i].annotations().push_back(number((double)sums[(size_t)i] / bed_file[i].length(), decimals));
}

BedFile Statistics::lowOrHighCoverage(const BedFile& bed_file, const std::string& bam_file, int cutoff, int min_mapq, int min_baseq, bool is_high, bool random_access, const std::string& ref_file)
{
	if (!random_access && !bed_file.isSorted()) NB_THROW(ArgumentException, "Input BED file has to be sorted for sweep algorithm!");
	if (!random_access && cutoff > 255) NB_THROW(ArgumentException, "Cutoff cannot be bigger than 255!");   // WorkerLowOrHighCoverage.cpp:149
	BedFile output;
	if (bed_file.count() == 0) return output;
	BamReader reader(bam_file, ref_file, true, bed_file);
	reader.requireIndex();
	std::vector<ngsqc_region> regions = unionRegions(bed_file, reader, true);
	ngsqc_depth_params p{}; p.min_mapq = min_mapq; p.min_baseq = min_baseq; p.regions = regions.data(); p.n_regions = (int64_t)regions.size();
	runDepthScan(reader, p);
	std::vector<ngsqc_region> lines = toRegions(bed_file, reader, true);
	int64_t n_runs = 0;
	reader.check(ngsqc_lowhigh_runs(reader.handle(), lines.data(), (int64_t)lines.size(), cutoff, is_high ? 1 : 0, random_access ? 0 : 1, nullptr, 0, &n_runs));
	std::vector<ngsqc_run> runs((size_t)std::max<int64_t>(n_runs, 1));
	if (n_runs) reader.check(ngsqc_lowhigh_runs(reader.handle(), lines.data(), (int64_t)lines.size(), cutoff, is_high ? 1 : 0, random_access ? 0 : 1, runs.data(), n_runs, &n_runs));
	for (int64_t i = 0; i < n_runs; ++i) { const BedLine& src = bed_file[runs[(size_t)i].line]; output.append(BedLine(src.chr(), runs[(size_t)i].start, runs[(size_t)i].end, src.annotations())); }
	output.merge(true, true, true);   // Statistics.cpp:2655
	return output;
}
BedFile Statistics::lowCoverage(const BedFile& bed_file, const std::string& bam_file, int cutoff, int min_mapq, int min_baseq, int, const std::string& ref_file, bool random_access, bool) { return lowOrHighCoverage(bed_file, bam_file, cutoff, min_mapq, min_baseq, false, random_access, ref_file); }
BedFile Statistics::highCoverage(const BedFile& bed_file, const std::string& bam_file, int cutoff, int min_mapq, int min_baseq, int, const std::string& ref_file, bool random_access, bool) { return lowOrHighCoverage(bed_file, bam_file, cutoff, min_mapq, min_baseq, true, random_access, ref_file); }

} // namespace ngsbits
