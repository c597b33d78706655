// BamInfo - drop-in for src/BamInfo/main.cpp:20-70 on the MI355X path: same flags and TSV output. Only the BGZF members of the BAM header, of the
// first records and (hg38) of the region behind the false-duplication mask go to the GPU (ngsqc_open_head / ngsqc_open_regions).
#include "Statistics.hpp"
#include <climits>
#include <cstdlib>
using namespace ngsbits;

class ConcreteTool : public ToolBase
{
public:
	ConcreteTool(int argc, char** argv) : ToolBase(argc, argv) {}
	void setup() override
	{
		setDescription("Basic BAM information.");
		addInfileList("in", "Input BAM/CRAM files.", false);
		addOutfile("out", "Output TSV file. If unset, writes to STDOUT.", true);
		addFlag("name", "Add filename only to output. The default is to add the canonical file path.");
		addInfile("ref", "Reference genome for CRAM support (mandatory if CRAM is used).", true);
		// --changelog (src/BamInfo/main.cpp)
		changeLog(2025, 9, 6, "First version.");
	}
	void main() override
	{
		std::string text = "#filename\tformat\tgenome_build\tgenome_masked\tgenome_contains_alt\tmapper\tpaired-end\n";
		const bool name = getFlag("name");
		for (const std::string& filename : getInfileList("in"))
		{
			BamReader reader(filename, getInfile("ref"), BamReader::Head{8});
			BamInfo info = reader.info();
			std::string shown = fileName(filename);
			if (!name) { char buf[PATH_MAX]; shown = realpath(filename.c_str(), buf) ? std::string(buf) : filename; }   // QFileInfo::canonicalFilePath
			text += shown + "\t" + info.file_format + "\t" + info.build + "\t" + (info.false_duplications_masked ? "yes" : "no") + "\t" + (info.contains_alt_chrs ? "yes" : "no") + "\t"
			        + trimmed(info.mapper + " " + info.mapper_version) + "\t" + (info.paired_end ? "yes" : "no") + "\n";
		}
		std::string out = getOutfile("out");
		if (out.empty()) fwrite(text.data(), 1, text.size(), stdout);
		else { FILE* f = fopen(out.c_str(), "wb"); if (!f) NB_THROW(FileAccessException, "Could not open file for writing: '" + out + "'!"); fwrite(text.data(), 1, text.size(), f); fclose(f); }
	}
};
int main(int argc, char** argv) { ConcreteTool tool(argc, argv); return tool.execute(); }
