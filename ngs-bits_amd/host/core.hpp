// Host layer above the C ABI (include/ngsqc.h), plain C++17 (the reference's Qt6 / cppCORE toolchain is absent in this
// image — SURVEY.md §8c). It mirrors, for the hot path only, the reference classes the four tools are written against:
//   Chromosome        src/cppNGS/Chromosome.{h,cpp}          BedLine/BedFile   src/cppNGS/BedFile.{h,cpp}
//   Histogram         cppCORE (un-vendored) QCCollection      src/cppNGS/QCCollection.{h,cpp}
//   FastaFileIndex    src/cppNGS/FastaFileIndex.cpp           ToolBase          cppCORE CLI conventions (doc/tools/*.md)
// Same names, argument meaning and error messages, so the tool mains read like the reference's.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <climits>
#include <cerrno>
#include <unordered_map>
#include <vector>

namespace ngsbits {

// ---- exceptions (names of the reference's exception classes; message text kept) ----
struct Exception : std::runtime_error { std::string type; Exception(const std::string& t, const std::string& m) : std::runtime_error(m), type(t) {} };
#define NB_THROW(TYPE, MSG) throw ::ngsbits::Exception(#TYPE, (MSG))

inline std::string trimmed(const std::string& s)
{
	size_t a = 0, b = s.size();
	while (a < b && isspace((unsigned char)s[a])) ++a;
	while (b > a && isspace((unsigned char)s[b - 1])) --b;
	return s.substr(a, b - a);
}
inline std::vector<std::string> split(const std::string& s, char sep)
{
	std::vector<std::string> f; size_t a = 0;
	while (true) { size_t b = s.find(sep, a); if (b == std::string::npos) { f.push_back(s.substr(a)); break; } f.push_back(s.substr(a, b - a)); a = b + 1; }
	return f;
}
inline std::string join(const std::vector<std::string>& v, const std::string& sep) { std::string o; for (size_t i = 0; i < v.size(); ++i) { if (i) o += sep; o += v[i]; } return o; }
inline std::string number(double v, int prec) { if (v != v) return "nan"; char b[64]; snprintf(b, sizeof(b), "%.*f", prec, v); return b; }  // QString::number(d,'f',prec); Qt prints a NaN (0 reads: 0 / 0) as "nan", printf as "-nan"
inline std::string fileName(const std::string& p) { size_t i = p.find_last_of('/'); return i == std::string::npos ? p : p.substr(i + 1); }   // QFileInfo::fileName
inline std::string baseName(const std::string& p) { std::string f = fileName(p); size_t i = f.find('.'); return i == std::string::npos ? f : f.substr(0, i); } // QFileInfo::baseName
std::string& defaultReferenceGenome();   // the genome of the settings (src/cppNGS/RefGenomeService.h), set by ToolBase before main()
inline bool fileExists(const std::string& p) { std::ifstream f(p); return (bool)f; }
// an index next to a BAM under one of the names sam_index_load tries (htslib hts_idx_check_local: <bam>.csi, <stem>.csi, <bam>.bai, <stem>.bai)
inline bool hasBamIndex(const std::string& bam)
{
	const size_t dot = bam.rfind('.'); const std::string stem = dot == std::string::npos ? bam : bam.substr(0, dot);
	if (bam.size() > 5 && bam.compare(bam.size() - 5, 5, ".cram") == 0) return fileExists(bam + ".crai") || fileExists(stem + ".crai");   // (a CRAM is indexed by a .crai)
	return fileExists(bam + ".csi") || fileExists(stem + ".csi") || fileExists(bam + ".bai") || fileExists(stem + ".bai");
}
inline std::string htmlEscaped(const std::string& s) // QString::toHtmlEscaped
{
	std::string o;
	for (char c : s) { if (c == '<') o += "&lt;"; else if (c == '>') o += "&gt;"; else if (c == '&') o += "&amp;"; else if (c == '"') o += "&quot;"; else o += c; }
	return o;
}

// ---- Chromosome (Chromosome.cpp:133-190) ----
class Chromosome
{
public:
	Chromosome() {}
	Chromosome(const std::string& s) : str_(trimmed(s)) { num_ = numericRepresentation(); }
	Chromosome(const char* s) : Chromosome(std::string(s)) {}
	bool operator<(const Chromosome& r) const { return num_ < r.num_; }
	bool operator>(const Chromosome& r) const { return num_ > r.num_; }
	bool operator==(const Chromosome& r) const { return num_ == r.num_; }
	bool operator!=(const Chromosome& r) const { return num_ != r.num_; }
	bool isValid() const { return num_ > 0; }
	bool isNonSpecial() const { return num_ > 0 && num_ < 1004; }
	bool isX() const { return num_ == 1001; }
	bool isY() const { return num_ == 1002; }
	const std::string& str() const { return str_; }
	int num() const { return num_; }
	std::string strNormalized(bool prepend_chr) const { return (prepend_chr ? "chr" : "") + normalized(); }
private:
	std::string str_; int num_ = 0;
	std::string normalized() const
	{
		std::string t; for (char c : str_) t.push_back((char)toupper((unsigned char)c));
		if (t.rfind("CHR", 0) == 0) t = t.substr(3);
		if (t == "M") t = "MT";
		return t;
	}
	int numericRepresentation() const
	{
		std::string t = normalized();
		if (t.empty()) return 0;
		if (t == "X") return 1001;
		if (t == "Y") return 1002;
		if (t == "MT") return 1003;
		if (t[0] != '0')
		{
			bool digits = t.size() <= 9; for (char c : t) if (!isdigit((unsigned char)c)) digits = false;
			if (digits) { long v = atol(t.c_str()); if (v > 0 && v <= 1000) return (int)v; }
		}
		static std::mutex m; static std::unordered_map<std::string, int> cache; static int next_num = 1004;
		std::lock_guard<std::mutex> g(m);
		auto it = cache.find(t); if (it == cache.end()) it = cache.emplace(t, next_num++).first;
		return it->second;
	}
};

// ---- BedLine / BedFile (BedFile.cpp) ----
class BedLine
{
public:
	BedLine() {}
	BedLine(const Chromosome& c, int s, int e, std::vector<std::string> a = {}) : chr_(c), start_(s), end_(e), annotations_(std::move(a)) {}
	const Chromosome& chr() const { return chr_; }
	int start() const { return start_; }
	int end() const { return end_; }
	void setStart(int s) { start_ = s; }
	void setEnd(int e) { end_ = e; }
	int length() const { return end_ - start_ + 1; }
	std::vector<std::string>& annotations() { return annotations_; }
	const std::vector<std::string>& annotations() const { return annotations_; }
	bool operator<(const BedLine& r) const { if (chr_ < r.chr_) return true; if (chr_ > r.chr_) return false; if (start_ == r.start_) return end_ < r.end_; return start_ < r.start_; }
	bool overlapsWith(int s, int e) const { return start_ <= e && s <= end_; }
	bool overlapsWith(const Chromosome& c, int s, int e) const { return chr_ == c && overlapsWith(s, e); }
	bool adjacentTo(const Chromosome& c, int s, int e) const { return chr_ == c && (start_ == e + 1 || end_ == s - 1); }
private:
	Chromosome chr_; int start_ = 0, end_ = -1; std::vector<std::string> annotations_;
};

class BedFile
{
public:
	BedFile() {}
	BedFile(const Chromosome& c, int s, int e) { append(BedLine(c, s, e)); }
	long long count() const { return (long long)lines_.size(); }
	const BedLine& operator[](long long i) const { return lines_[(size_t)i]; }
	BedLine& operator[](long long i) { return lines_[(size_t)i]; }
	const std::vector<std::string>& headers() const { return headers_; }
	void appendHeader(const std::string& h) { headers_.push_back(h); }
	void clearHeaders() { headers_.clear(); }
	void clearAnnotations() { for (auto& l : lines_) l.annotations().clear(); }
	void clear() { lines_.clear(); headers_.clear(); }
	void append(const BedLine& l)
	{
		if (!l.chr().isValid()) NB_THROW(ArgumentException, "Invalid BED line chromosome - empty string!");
		if (l.start() < 1 || l.end() < 1 || l.start() > l.end()) NB_THROW(ArgumentException, "Invalid BED line range '" + std::to_string(l.start()) + "' to '" + std::to_string(l.end()) + "'!");
		lines_.push_back(l);
	}
	void add(const BedFile& o) { for (auto& l : o.lines_) lines_.push_back(l); }
	long long baseCount() const { long long o = 0; for (auto& l : lines_) o += l.length(); return o; }
	std::vector<Chromosome> chromosomes() const { std::vector<Chromosome> o; for (auto& l : lines_) { bool f = false; for (auto& c : o) if (c == l.chr()) f = true; if (!f) o.push_back(l.chr()); } return o; }

	void loadStream(std::istream& in, bool read_annotations = true)
	{
		clear(); std::string line;
		while (std::getline(in, line))
		{
			while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
			if (line.empty()) continue;
			if (line[0] == '#' || line.rfind("track ", 0) == 0 || line.rfind("browser ", 0) == 0 || line.rfind("Chromosome\tStart\tEnd", 0) == 0) { headers_.push_back(line); continue; }
			std::vector<std::string> f = split(line, '\t');
			if (f.size() < 3) NB_THROW(FileParseException, "BED file line with less than three fields found: '" + trimmed(line) + "'");
			// QByteArray::toInt(&ok) (BedFile.cpp:157-160): base 10, white space around the number ignored, not ok outside the range of int
			auto to_int = [](const std::string& t, long& v) -> bool {
				char* e = nullptr; errno = 0; v = strtol(t.c_str(), &e, 10);
				if (e == t.c_str() || errno == ERANGE || v < INT_MIN || v > INT_MAX) return false;
				while (*e == ' ' || *e == '\t' || *e == '\n' || *e == '\v' || *e == '\f' || *e == '\r') ++e;
				return *e == 0;
			};
			long s = 0, e = 0;
			if (!to_int(f[1], s)) NB_THROW(FileParseException, "BED file line with invalid starts position found: '" + trimmed(line) + "'");
			if (!to_int(f[2], e)) NB_THROW(FileParseException, "BED file line with invalid end position found: '" + trimmed(line) + "'");
			std::vector<std::string> annos; if (read_annotations) annos.assign(f.begin() + 3, f.end());
			append(BedLine(Chromosome(f[0]), (int)s + 1, (int)e, annos));
		}
	}
	void load(const std::string& filename, bool stdin_if_empty = true, bool read_annotations = true)
	{
		if (filename.empty() && stdin_if_empty) { loadStream(std::cin, read_annotations); return; }
		std::ifstream f(filename, std::ios::binary);
		if (!f) NB_THROW(FileAccessException, "Could not open file for reading: '" + filename + "'!");
		loadStream(f, read_annotations);
	}
	std::string toText() const
	{
		std::string o;
		for (auto& h : headers_) o += trimmed(h) + "\n";
		for (auto& l : lines_) { o += l.chr().str() + "\t" + std::to_string(l.start() - 1) + "\t" + std::to_string(l.end()); for (auto& a : l.annotations()) o += "\t" + a; o += "\n"; }
		return o;
	}
	void store(const std::string& filename, bool stdout_if_empty = true) const
	{
		std::string t = toText();
		if (filename.empty() && stdout_if_empty) { fwrite(t.data(), 1, t.size(), stdout); return; }
		FILE* f = fopen(filename.c_str(), "wb");
		if (!f) NB_THROW(FileAccessException, "Could not open file for writing: '" + filename + "'!");
		fwrite(t.data(), 1, t.size(), f); fclose(f);
	}
	bool isSorted() const { for (size_t i = 1; i < lines_.size(); ++i) if (lines_[i] < lines_[i - 1]) return false; return true; }
	bool isMergedAndSorted() const
	{
		for (size_t i = 1; i < lines_.size(); ++i)
		{
			if (lines_[i] < lines_[i - 1]) return false;
			if (lines_[i - 1].overlapsWith(lines_[i].chr(), lines_[i].start(), lines_[i].end())) return false;
		}
		return true;
	}
	void sort() { std::stable_sort(lines_.begin(), lines_.end()); }
	void merge(bool merge_back_to_back = true, bool merge_names = false, bool merged_names_unique = false)
	{
		if (lines_.empty()) return;
		if (!merge_names) clearAnnotations();
		for (auto& l : lines_) if (merge_names) { std::string name = l.annotations().empty() ? "" : l.annotations()[0]; l.annotations().assign(1, name); }
		if (!isSorted()) sort();
		BedLine next = lines_[0]; size_t out = 0;
		for (size_t i = 1; i < lines_.size(); ++i)
		{
			const BedLine line = lines_[i];
			if (next.overlapsWith(line.chr(), line.start(), line.end()) || (merge_back_to_back && next.adjacentTo(line.chr(), line.start(), line.end())))
			{
				if (line.end() > next.end()) next.setEnd(line.end());
				if (merge_names)
				{
					const std::string& a = line.annotations()[0];
					if (!merged_names_unique || std::find(next.annotations().begin(), next.annotations().end(), a) == next.annotations().end()) next.annotations().push_back(a);
				}
			}
			else
			{
				lines_[out] = next;
				if (merge_names) lines_[out].annotations().assign(1, join(next.annotations(), ","));
				++out; next = line;
			}
		}
		lines_[out] = next;
		if (merge_names) lines_[out].annotations().assign(1, join(next.annotations(), ","));
		lines_.resize(out + 1);
	}
	void chunk(int chunk_size)
	{
		std::vector<BedLine> nl; nl.reserve(lines_.size());
		for (auto& line : lines_)
		{
			if (line.length() > chunk_size)
			{
				double length = line.length(); int n = (int)floor(length / chunk_size);
				if (fabs(chunk_size - (length / n)) > fabs(chunk_size - (length / (n + 1)))) n += 1;
				std::vector<int> sizes((size_t)n, chunk_size);
				int rest = line.length() - n * chunk_size, cur = 0;
				while (rest != 0) { int sign = rest > 0 ? 1 : -1; sizes[cur] += sign; rest -= sign; ++cur; if (cur == n) cur = 0; }
				int start = line.start(); BedLine x = line;
				for (int i = 0; i < n; ++i) { int end = start + sizes[i] - 1; x.setStart(start); x.setEnd(end); nl.push_back(x); start = end + 1; }
			}
			else nl.push_back(line);
		}
		lines_.swap(nl);
	}
private:
	std::vector<BedLine> lines_; std::vector<std::string> headers_;
};

// ---- Histogram (cppCORE; restated from its use) ----
class Histogram
{
public:
	Histogram(double mn, double mx, double bin) : min_(mn), max_(mx), bins_((size_t)ceil((mx - mn) / bin), 0.0) {}
	int binCount() const { return (int)bins_.size(); }
	int binIndex(double v) const
	{
		if (v < min_ || v > max_) NB_THROW(StatisticsException, "Requested position '" + number(v, 6) + "' not in range (" + number(min_, 6) + "-" + number(max_, 6) + ")!");
		int i = (int)floor((v - min_) / (max_ - min_) * bins_.size());
		return std::min(std::max(0, i), (int)bins_.size() - 1);
	}
	void inc(double v, bool ignore_bounds, double by = 1.0) { if (ignore_bounds) v = std::min(std::max(v, min_), max_); bins_[binIndex(v)] += by; sum_ += by; }
	double binValue(int i, bool percentage = false) const { return percentage ? 100.0 * bins_[i] / sum_ : bins_[i]; }
	double binSum() const { return sum_; }
	std::vector<double> xCoords() const { std::vector<double> x; double w = (max_ - min_) / bins_.size(); for (size_t i = 0; i < bins_.size(); ++i) x.push_back(min_ + (i + 0.5) * w); return x; }
	std::vector<double> yCoords(bool percentage) const { std::vector<double> y; for (size_t i = 0; i < bins_.size(); ++i) y.push_back(binValue((int)i, percentage)); return y; }
private:
	double min_, max_; std::vector<double> bins_; double sum_ = 0;
};

// ---- FastaFileIndex (FastaFileIndex.cpp:10-154) ----
class FastaFileIndex
{
public:
	explicit FastaFileIndex(const std::string& fasta_file) : name_(fasta_file)
	{
		f_ = fopen(fasta_file.c_str(), "rb");
		if (!f_) NB_THROW(FileAccessException, "Could not open FASTA file '" + fasta_file + "' for reading!");
		std::ifstream fai(fasta_file + ".fai");
		if (!fai) NB_THROW(FileAccessException, "Could not open file for reading: '" + fasta_file + ".fai'!");
		std::string line; int n = 0;
		while (std::getline(fai, line))
		{
			++n; std::vector<std::string> fl = split(line, '\t');
			if (fl.size() != 5) NB_THROW(FileParseException, "Malformed FASTA index line " + std::to_string(n) + " in file '" + fasta_file + ".fai'!");
			idx_[Chromosome(fl[0]).num()] = Entry{atoi(fl[1].c_str()), atoll(fl[2].c_str()), atoi(fl[3].c_str())};
			const Entry& en = idx_[Chromosome(fl[0]).num()];
			if (en.length < 0 || en.offset < 0 || en.line_blen <= 0) NB_THROW(FileParseException, "Malformed FASTA index line " + std::to_string(n) + " in file '" + fasta_file + ".fai'!");   // (a line length of 0 would divide by zero in seq())
		}
		if (idx_.empty()) NB_THROW(FileParseException, "Empty FAI file for " + fasta_file + "'!");
	}
	~FastaFileIndex() { if (f_) fclose(f_); }
	std::string seq(const Chromosome& chr, int start, int length, bool to_upper = true) const
	{
		start -= 1;
		const Entry& e = index(chr);
		if (start > e.length) NB_THROW(ProgrammingException, "FastaFileIndex::seq: Invalid start position");
		if (start + length > e.length) length = std::min(length, e.length - start);
		int nl_before = start > 0 ? (start - 1) / e.line_blen : 0;   // kept as in FastaFileIndex.cpp:96
		long long pos = e.offset + nl_before + start;
		int seqlen = length + ((start + length - 1) / e.line_blen - nl_before);
		std::string s = raw(pos, seqlen);
		if (to_upper) for (auto& c : s) c = (char)toupper((unsigned char)c);
		return s;
	}
	int n(const Chromosome& chr) const
	{
		auto c = ncache_.find(chr.num()); if (c != ncache_.end()) return c->second;
		const Entry& e = index(chr);
		std::string s = raw(e.offset, e.length / e.line_blen + e.length);
		int o = 0; for (char ch : s) if (ch == 'N' || ch == 'n') ++o;
		return ncache_[chr.num()] = o;
	}
private:
	struct Entry { int length; long long offset; int line_blen; };
	const Entry& index(const Chromosome& chr) const { auto it = idx_.find(chr.num()); if (it == idx_.end()) NB_THROW(ArgumentException, "Unknown FASTA index chromosome '" + chr.str() + "' requested!"); return it->second; }
	std::string raw(long long pos, int n) const
	{
		std::string s((size_t)std::max(n, 0), '\0'); fseeko(f_, pos, SEEK_SET);
		size_t got = fread(&s[0], 1, s.size(), f_); s.resize(got);
		s.erase(std::remove(s.begin(), s.end(), '\n'), s.end());
		return s;
	}
	std::string name_; FILE* f_ = nullptr; std::map<int, Entry> idx_; mutable std::map<int, int> ncache_;
};
inline double gcContent(const std::string& s) // Sequence.cpp:86-101
{
	int gc = 0, at = 0; for (char b : s) { if (b == 'G' || b == 'C') ++gc; else if (b == 'A' || b == 'T') ++at; }
	if (gc + at == 0) return std::numeric_limits<double>::quiet_NaN();
	return (double)gc / (gc + at);
}

// ---- QCValue / QCCollection (QCCollection.cpp:121-384) ----
enum class QCValueType { DOUBLE, STRING, IMAGE };
struct QCValue
{
	std::string name, description, accession; QCValueType type = QCValueType::STRING; double d = 0; std::string s; // s: string value or base64 PNG
	std::string toString(int prec = 2) const { return type == QCValueType::DOUBLE ? number(d, prec) : s; }
	double asDouble() const { if (type != QCValueType::DOUBLE) NB_THROW(TypeConversionException, "QCValue '" + name + "' requested as double, but has different type!"); return d; }
};

std::string resourceDir();   // ngs-bits_amd/resources (next to the tool binaries' parent)

class QCCollection
{
public:
	int count() const { return (int)values_.size(); }
	const QCValue& operator[](int i) const { return values_[(size_t)i]; }
	void insert(const QCValue& v) { for (auto& x : values_) if (x.name == v.name) { x = v; return; } values_.push_back(v); }
	void insert(const QCCollection& c) { for (auto& v : c.values_) insert(v); }
	const QCValue& value(const std::string& key, bool by_accession) const
	{
		for (auto& v : values_) if ((by_accession ? v.accession : v.name) == key) return v;
		NB_THROW(ArgumentException, "QC value with name/accession '" + key + "' not found in QC collection.");
	}
	void appendToStringList(std::vector<std::string>& list) const { for (auto& v : values_) if (v.type != QCValueType::IMAGE) list.push_back(v.name + ": " + v.toString()); }
	void storeToQCML(const std::string& filename, const std::vector<std::string>& source_files, const std::string& parameters, const std::string& app_name, const std::string& app_version) const;
private:
	std::vector<QCValue> values_;
};

// qcML ontology subset (resources/qcml_terms.tsv): accession -> (name, definition). Statistics::addQcValue checks both.
struct OntologyTerm { std::string name, definition; };
const std::map<std::string, OntologyTerm>& qcmlTerms();

// ---- ToolBase: the reference's CLI conventions (single-dash named parameters, -flag, --help/--version) ----
class ToolBase
{
public:
	ToolBase(int argc, char** argv) { for (int i = 0; i < argc; ++i) args_.push_back(argv[i]); }
	virtual ~ToolBase() {}
	virtual void setup() = 0;
	virtual void main() = 0;
	int execute();
protected:
	void setDescription(const std::string& d) { description_ = d; }
	void setExtendedDescription(const std::vector<std::string>& d) { ext_ = d; }
	void addInfile(const std::string& n, const std::string& d, bool optional, bool = true) { add(n, "infile", d, optional, ""); }
	void addInfileList(const std::string& n, const std::string& d, bool optional) { add(n, "infilelist", d, optional, ""); }
	void addOutfile(const std::string& n, const std::string& d, bool optional) { add(n, "outfile", d, optional, ""); }
	void addInt(const std::string& n, const std::string& d, bool optional, int def = 0) { add(n, "int", d, optional, std::to_string(def)); }
	void addFloat(const std::string& n, const std::string& d, bool optional, double def = 0.0) { add(n, "float", d, optional, number(def, 6)); }
	void addFlag(const std::string& n, const std::string& d) { add(n, "flag", d, true, ""); }
	void addEnum(const std::string& n, const std::string& d, bool optional, const std::vector<std::string>& values, const std::string& def) { add(n, "enum", d, optional, def); params_.back().values = values; }
	void changeLog(int y, int m, int d, const std::string& text) { char b[16]; snprintf(b, sizeof(b), "%04d-%02d-%02d", y, m, d); changelog_.push_back(std::string(b) + " " + text); }
	std::string getInfile(const std::string& n) const { return get(n).value; }
	std::vector<std::string> getInfileList(const std::string& n) const { return get(n).list; }
	std::string getOutfile(const std::string& n) const { return get(n).value; }
	int getInt(const std::string& n) const { return atoi(get(n).value.c_str()); }
	double getFloat(const std::string& n) const { return atof(get(n).value.c_str()); }
	bool getFlag(const std::string& n) const { return get(n).set; }
	std::string getEnum(const std::string& n) const { return get(n).value; }
	std::string appName() const { return fileName(args_.empty() ? "" : args_[0]); }
	std::string settingsString(const std::string& key) const;   // <bin dir>/settings.ini "key = value"
	static std::string version() { return "ngsqc-mi355x-0.1"; }
private:
	struct Param { std::string name, type, desc; bool optional; std::string value; std::vector<std::string> list; bool set = false; std::vector<std::string> values; };
	void add(const std::string& n, const std::string& t, const std::string& d, bool optional, const std::string& def) { Param p; p.name = n; p.type = t; p.desc = d; p.optional = optional; p.value = def; params_.push_back(p); }
	const Param& get(const std::string& n) const { for (auto& p : params_) if (p.name == n) return p; NB_THROW(ProgrammingException, "Unknown parameter '" + n + "'"); }
	void parse();
	void printHelp() const;
	void printChangelog() const;
	void storeTDX() const;
	std::vector<std::string> changelog_; std::string settings_override_;
	std::vector<std::string> args_; std::string description_; std::vector<std::string> ext_; std::vector<Param> params_;
};

} // namespace ngsbits
