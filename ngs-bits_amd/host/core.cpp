// Implementation of the host-layer helpers declared in core.hpp (resource lookup, qcML writer, CLI conventions).
#include "core.hpp"
#include <ctime>
#include <unistd.h>

namespace ngsbits {

static std::string exeDir()
{
	char buf[4096]; ssize_t n = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
	if (n <= 0) return ".";
	buf[n] = 0; std::string p(buf); size_t i = p.find_last_of('/');
	return i == std::string::npos ? "." : p.substr(0, i);
}

std::string resourceDir()
{
	const char* e = getenv("NGSQC_RESOURCES");
	if (e && *e) return e;
	std::string d = exeDir();   // <repo>/ngs-bits_amd/bin -> <repo>/ngs-bits_amd/resources
	for (const char* rel : {"/../resources", "/resources", "/../../ngs-bits_amd/resources"}) if (fileExists(d + rel + "/qcml_terms.tsv")) return d + rel;
	return d + "/../resources";
}

const std::map<std::string, OntologyTerm>& qcmlTerms()
{
	static std::map<std::string, OntologyTerm> terms;
	static std::once_flag once;
	std::call_once(once, [] {
		std::ifstream f(resourceDir() + "/qcml_terms.tsv");
		if (!f) NB_THROW(FileAccessException, "Could not open qcML term table '" + resourceDir() + "/qcml_terms.tsv'!");
		std::string line;
		while (std::getline(f, line)) { auto p = split(line, '\t'); if (p.size() >= 3) terms[p[0]] = OntologyTerm{p[1], p[2]}; }
	});
	return terms;
}

static std::string pad4(int i) { char b[16]; snprintf(b, sizeof(b), "%04d", i); return b; }

// QCCollection::storeToQCML (QCCollection.cpp:200-339). Layout facts: SURVEY.md §8 a16b.
void QCCollection::storeToQCML(const std::string& filename, const std::vector<std::string>& source_files, const std::string& parameters, const std::string& app_name, const std::string& app_version) const
{
	std::string o;
	o += "<?xml version=\"1.0\" encoding=\"ISO-8859-1\"?>\n";
	o += "<?xml-stylesheet type=\"text/xml\" href=\"#stylesheet\"?>\n";
	o += "<!DOCTYPE catelog [\n  <!ATTLIST xsl:stylesheet\n  id  ID  #REQUIRED>\n  ]>\n";
	o += "<qcML version=\"0.0.8\" xmlns=\"http://www.prime-xs.eu/ms/qcml\" >\n";
	o += "  <runQuality ID=\"rq0001\">\n";
	char date[64]; time_t t = time(nullptr); struct tm tmv; localtime_r(&t, &tmv); strftime(date, sizeof(date), "%Y-%m-%dT%H:%M:%S", &tmv);
	o += "    <metaDataParameter ID=\"md0001\" name=\"creation software\" value=\"" + app_name + " " + app_version + "\" cvRef=\"QC\" accession=\"QC:1000002\"/>\n";
	o += "    <metaDataParameter ID=\"md0002\" name=\"creation software parameters\" value=\"" + htmlEscaped(parameters) + "\" cvRef=\"QC\" accession=\"QC:1000003\"/>\n";
	o += std::string("    <metaDataParameter ID=\"md0003\" name=\"creation date\" value=\"") + date + "\" cvRef=\"QC\" accession=\"QC:1000004\"/>\n";
	int idx = 4;
	for (auto& sf : source_files) { o += "    <metaDataParameter ID=\"md" + pad4(idx) + "\" name=\"source file\" value=\"" + fileName(sf) + "\" cvRef=\"QC\" accession=\"QC:1000005\"/>\n"; ++idx; }
	for (int i = 0; i < count(); ++i)
	{
		const QCValue& v = values_[(size_t)i];
		if (v.type == QCValueType::IMAGE) continue;
		o += "    <qualityParameter ID=\"qp" + pad4(i + 1) + "\" name=\"" + v.name + "\" description=\"" + htmlEscaped(v.description) + "\" value=\"" + v.toString() + "\" cvRef=\"QC\" accession=\"" + v.accession + "\"/>\n";
	}
	for (int i = 0; i < count(); ++i)
	{
		const QCValue& v = values_[(size_t)i];
		if (v.type != QCValueType::IMAGE) continue;
		o += "    <attachment ID=\"qp" + pad4(i + 1) + "\" name=\"" + v.name + "\" description=\"" + htmlEscaped(v.description) + "\" cvRef=\"QC\" accession=\"" + v.accession + "\">\n";
		o += "      <binary>" + v.s + "</binary>\n";
		o += "    </attachment>\n";
	}
	o += "  </runQuality>\n";
	{
		std::ifstream f(resourceDir() + "/qcml_tail.txt", std::ios::binary);
		if (!f) NB_THROW(FileAccessException, "Could not open qcML stylesheet block '" + resourceDir() + "/qcml_tail.txt'!");
		std::stringstream ss; ss << f.rdbuf(); o += ss.str();
	}
	if (filename.empty()) { fwrite(o.data(), 1, o.size(), stdout); return; }
	FILE* f = fopen(filename.c_str(), "wb");
	if (!f) NB_THROW(FileAccessException, "Could not open file for writing: '" + filename + "'!");
	fwrite(o.data(), 1, o.size(), f); fclose(f);
}

std::string& defaultReferenceGenome() { static std::string g; return g; }

std::string ToolBase::settingsString(const std::string& key) const
{
	// --settings <file>: that file only (doc/tools/*.md "Settings override file (no other settings files are used)")
	for (const std::string& p : settings_override_.empty() ? std::vector<std::string>{exeDir() + "/settings.ini", exeDir() + "/../settings.ini"} : std::vector<std::string>{settings_override_})
	{
		std::ifstream f(p); if (!f) continue;
		std::string line;
		while (std::getline(f, line))
		{
			size_t eq = line.find('='); if (eq == std::string::npos) continue;
			if (trimmed(line.substr(0, eq)) == key) return trimmed(line.substr(eq + 1));
		}
	}
	return "";
}

void ToolBase::printHelp() const
{
	printf("%s (%s)\n\n%s\n", appName().c_str(), version().c_str(), description_.c_str());
	for (auto& e : ext_) printf("%s\n", e.c_str());
	printf("\nMandatory parameters:\n");
	for (auto& p : params_) if (!p.optional) printf("  -%s <%s>\t%s\n", p.name.c_str(), p.type.c_str(), p.desc.c_str());
	printf("\nOptional parameters:\n");
	for (auto& p : params_) if (p.optional)
	{
		if (p.type == "flag") printf("  -%s\t%s\n\t\tDefault value: 'false'\n", p.name.c_str(), p.desc.c_str());
		else printf("  -%s <%s>\t%s\n\t\tDefault value: '%s'\n", p.name.c_str(), p.type.c_str(), p.desc.c_str(), p.value.c_str());
	}
	printf("\nSpecial parameters:\n  --help\tShows this help and exits.\n  --version\tPrints version and exits.\n  --changelog\tPrints changeloge and exits.\n"
	       "  --tdx\tWrites a Tool Definition Xml file. The file name is the application name with the suffix '.tdx'.\n  --settings [file]\tSettings override file (no other settings files are used).\n");
}

// doc/tools/<Tool>.md "### <Tool> changelog": the tool's name and version, an empty line, one line per entry
void ToolBase::printChangelog() const
{
	printf("%s %s\n\n", appName().c_str(), version().c_str());
	for (auto& c : changelog_) printf("%s\n", c.c_str());
}

// Tool Definition Xml (cppCORE ToolBase::storeTDXml; the source is not in the tree - the element names follow the TDX schema of ngs-bits, src/cppCORE is an empty submodule)
void ToolBase::storeTDX() const
{
	auto esc = [](const std::string& t) { std::string o; for (char c : t) { if (c == '&') o += "&amp;"; else if (c == '<') o += "&lt;"; else if (c == '>') o += "&gt;"; else if (c == '"') o += "&quot;"; else o += c; } return o; };
	const std::string fn = appName() + ".tdx";
	FILE* f = fopen(fn.c_str(), "wb"); if (!f) NB_THROW(FileAccessException, "Could not open file for writing: '" + fn + "'!");
	fprintf(f, "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<TDX version=\"1\">\n  <Tool name=\"%s\" version=\"%s\">\n    <Description>%s</Description>\n", esc(appName()).c_str(), esc(version()).c_str(), esc(description_).c_str());
	if (!ext_.empty()) { std::string e; for (auto& l : ext_) e += (e.empty() ? "" : "\n") + l; fprintf(f, "    <ExtendedDescription>%s</ExtendedDescription>\n", esc(e).c_str()); }
	for (auto& p : params_)
	{
		const char* tag = p.type == "infile" ? "Infile" : p.type == "infilelist" ? "InfileList" : p.type == "outfile" ? "Outfile" : p.type == "int" ? "Int" : p.type == "float" ? "Float" : p.type == "enum" ? "Enum" : p.type == "flag" ? "Flag" : "String";
		fprintf(f, "    <%s name=\"%s\">\n      <Description>%s</Description>\n", tag, esc(p.name).c_str(), esc(p.desc).c_str());
		if (p.type != "flag") { if (p.optional) fprintf(f, "      <Optional defaultValue=\"%s\" />\n", esc(p.value).c_str()); }
		for (auto& v : p.values) fprintf(f, "      <Value>%s</Value>\n", esc(v).c_str());
		fprintf(f, "    </%s>\n", tag);
	}
	fprintf(f, "  </Tool>\n</TDX>\n");
	fclose(f);
}

void ToolBase::parse()
{
	std::vector<bool> given(params_.size(), false);
	for (size_t i = 1; i < args_.size(); ++i)
	{
		const std::string& a = args_[i];
		if (a.size() < 2 || a[0] != '-') NB_THROW(CommandLineParsingException, "Trailing parameter '" + a + "' given.");
		std::string name = a.substr(1);
		size_t k = params_.size();
		for (size_t j = 0; j < params_.size(); ++j) if (params_[j].name == name) k = j;
		if (k == params_.size()) NB_THROW(CommandLineParsingException, "Unknown parameter '" + name + "' given.");
		if (given[k]) NB_THROW(CommandLineParsingException, "Parameter '" + name + "' given more than once.");
		given[k] = true;
		Param& p = params_[k];
		if (p.type == "flag") { p.set = true; continue; }
		std::vector<std::string> vals;
		while (i + 1 < args_.size() && !(args_[i + 1].size() >= 2 && args_[i + 1][0] == '-' && !isdigit((unsigned char)args_[i + 1][1]))) vals.push_back(args_[++i]);
		if (p.type == "infilelist")
		{
			if (vals.empty()) NB_THROW(CommandLineParsingException, "Parameter '" + name + "' given without value.");
			for (auto& v : vals) if (!fileExists(v)) NB_THROW(CommandLineParsingException, "Input file '" + v + "' given for parameter '" + name + "' does not exist.");
			p.list = vals; p.set = true; continue;
		}
		if (vals.size() != 1) NB_THROW(CommandLineParsingException, vals.empty() ? "Parameter '" + name + "' given without value." : "Parameter '" + name + "' given with more than one value.");
		if (p.type == "int") { char* e; strtol(vals[0].c_str(), &e, 10); if (vals[0].empty() || *e) NB_THROW(CommandLineParsingException, "Value '" + vals[0] + "' given for parameter '" + name + "' cannot be converted to integer."); }
		if (p.type == "float") { char* e; strtod(vals[0].c_str(), &e); if (vals[0].empty() || *e) NB_THROW(CommandLineParsingException, "Value '" + vals[0] + "' given for parameter '" + name + "' cannot be converted to float."); }
		if (p.type == "enum" && std::find(p.values.begin(), p.values.end(), vals[0]) == p.values.end()) NB_THROW(CommandLineParsingException, "Value '" + vals[0] + "' given for parameter '" + name + "' is not valid. Valid are: '" + join(p.values, ",") + "'.");
		if ((p.type == "infile" || p.type == "infilelist") && !vals[0].empty() && !fileExists(vals[0])) NB_THROW(CommandLineParsingException, "Input file '" + vals[0] + "' given for parameter '" + name + "' does not exist.");
		p.value = vals[0]; p.set = true;
	}
	for (size_t j = 0; j < params_.size(); ++j) if (!params_[j].optional && !given[j]) NB_THROW(CommandLineParsingException, "Mandatory parameter '" + params_[j].name + "' not given.");
}

// NGSQC_TIMING: the age of the process (from the kernel's start time of the process: exec, the dynamic loader and the static initialisers lie in front of main)
static void process_age_stamp(const char* what)
{
	if (!getenv("NGSQC_TIMING")) return;
	double up = 0; unsigned long long start_ticks = 0;
	if (FILE* f = fopen("/proc/uptime", "r")) { if (fscanf(f, "%lf", &up) != 1) up = 0; fclose(f); }
	if (FILE* f = fopen("/proc/self/stat", "r"))
	{
		char buf[2048]; const size_t n = fread(buf, 1, sizeof(buf) - 1, f); buf[n] = 0; fclose(f);
		if (const char* p = strrchr(buf, ')')) { int field = 2; for (const char* q = p + 1; *q; ++q) if (*q == ' ' && ++field == 22) { start_ticks = strtoull(q + 1, nullptr, 10); break; } }
	}
	const long hz = sysconf(_SC_CLK_TCK);
	if (up > 0 && start_ticks && hz > 0) fprintf(stderr, "[ngsqc] process age %.2f s at %s\n", up - (double)start_ticks / (double)hz, what);
}

int ToolBase::execute()
{
	try
	{
		setup();
		for (size_t i = 1; i < args_.size(); ++i)
		{
			if (args_[i] == "--help") { printHelp(); return 0; }
			if (args_[i] == "--version") { printf("%s %s\n", appName().c_str(), version().c_str()); return 0; }
			if (args_[i] == "--changelog") { printChangelog(); return 0; }
			if (args_[i] == "--tdx") { storeTDX(); return 0; }
		}
		// --settings <file> is taken out of the argument list before the tool's own parameters are parsed
		for (size_t i = 1; i < args_.size(); ++i)
			if (args_[i] == "--settings")
			{
				if (i + 1 >= args_.size() || (args_[i + 1].size() > 1 && args_[i + 1][0] == '-')) NB_THROW(CommandLineParsingException, "Parameter '--settings' given without value.");
				if (!fileExists(args_[i + 1])) NB_THROW(CommandLineParsingException, "Settings override file '" + args_[i + 1] + "' does not exist.");
				settings_override_ = args_[i + 1]; args_.erase(args_.begin() + (long)i, args_.begin() + (long)i + 2); --i;
			}
		parse();
		process_age_stamp("main (the loader has mapped the HIP runtime and this tool)");
		defaultReferenceGenome() = settingsString("reference_genome");   // (RefGenomeService::getReferenceGenome: what a BamReader without an explicit genome opens a CRAM with)
		main();
		// The outputs are written and closed. Leaving through exit() would run the static destructors and the HIP runtime's teardown, which gives tens of GB of
		// device memory back page by page (0.9 s for the buffers of a 60 GB BAM, profiles/r03_tool_probe.txt); the driver reclaims them at once when the process is gone.
		process_age_stamp("exit (outputs written and closed)");
		fflush(stdout); fflush(stderr);
		if (!getenv("NGSQC_SLOW_EXIT")) _exit(0);
		return 0;
	}
	catch (Exception& e)
	{
		if (e.type == "CommandLineParsingException") fprintf(stderr, "%s %s\nCommand line parsing exception: %s\nCall this tool with the argument '--help' for help.\n", appName().c_str(), version().c_str(), e.what());
		else fprintf(stderr, "%s %s\n%s: %s\n", appName().c_str(), version().c_str(), e.type.c_str(), e.what());
		return 1;
	}
	catch (std::exception& e) { fprintf(stderr, "%s %s\nException: %s\n", appName().c_str(), version().c_str(), e.what()); return 1; }
}

} // namespace ngsbits
