// SampleGender - drop-in for src/SampleGender/main.cpp on the MI355X path: same flags, defaults and TSV output. The three methods reuse
// the GPU passes of the mapping QC path: xy = the chrX / chrY counters of the scan (Statistics::yxRatio), hetx = the site
// pileup of the known chrX SNVs, sry = the per-base depth of the SRY gene (Statistics::avgCoverage).
#include "Statistics.hpp"
using namespace ngsbits;

class ConcreteTool : public ToolBase
{
public:
	ConcreteTool(int argc, char** argv) : ToolBase(argc, argv) {}
	void setup() override
	{
		setDescription("Determines the gender of a sample from the BAM/CRAM file.");
		addInfileList("in", "Input BAM/CRAM file(s).", false);
		addOutfile("out", "Output TSV file - one line per input BAM/CRAM file. If unset, writes to STDOUT.", true);
		addEnum("method", "Method selection: Read distribution on X and Y chromosome (xy), fraction of heterozygous variants on X chromosome (hetx), or coverage of SRY gene (sry).", false, {"xy", "hetx", "sry"}, "");
		addFloat("max_female", "Maximum Y/X ratio for female (method xy).", true, 0.06);
		addFloat("min_male", "Minimum Y/X ratio for male (method xy).", true, 0.09);
		addFloat("min_female", "Minimum heterozygous SNP fraction for female (method hetx).", true, 0.25);
		addFloat("max_male", "Maximum heterozygous SNP fraction for male (method hetx).", true, 0.05);
		addFloat("sry_cov", "Minimum average coverage of SRY gene for males (method sry).", true, 20.0);
		addEnum("build", "Genome build used to generate the input (methods hetx and sry).", true, {"hg19", "hg38"}, "hg38");
		addInfile("ref", "Reference genome for CRAM support (mandatory if CRAM is used).", true);
		addFlag("long_read", "Support long reads (> 1kb) and uses single-end reads for gender calculation.");
		// --changelog (src/SampleGender/main.cpp)
		changeLog(2024, 2, 29, "Added parameter to include single-end reads (long-read).");
		changeLog(2022, 8, 5, "Ignoring duplicate, secondary and supplementary alignments in methods 'xy' and 'sry' now.");
		changeLog(2020, 11, 27, "Added CRAM support.");
		changeLog(2018, 7, 13, "Change of output to TSV format for batch support.");
		changeLog(2018, 7, 11, "Added build switch for hg38 support.");
	}
	void main() override
	{
		std::vector<std::string> in = getInfileList("in");
		std::string method = getEnum("method"), build = getEnum("build");
		std::string text; bool print_header = true;
		for (const std::string& bam : in)
		{
			GenderEstimate estimate;
			if (method == "xy") estimate = Statistics::genderXY(bam, getFloat("max_female"), getFloat("min_male"), getInfile("ref"));
			else if (method == "hetx") estimate = Statistics::genderHetX(build, bam, getFloat("max_male"), getFloat("min_female"), getInfile("ref"), getFlag("long_read"));
			else estimate = Statistics::genderSRY(build, bam, getFloat("sry_cov"), getInfile("ref"));
			if (print_header)
			{
				text += "#file\tgender";
				for (auto& info : estimate.add_info) text += "\t" + info.first;
				text += "\n"; print_header = false;
			}
			text += fileName(bam) + "\t" + estimate.gender;
			for (auto& info : estimate.add_info) text += "\t" + info.second;
			text += "\n";
		}
		std::string out = getOutfile("out");
		if (out.empty()) fwrite(text.data(), 1, text.size(), stdout);
		else { FILE* f = fopen(out.c_str(), "wb"); if (!f) NB_THROW(FileAccessException, "Could not open file for writing: '" + out + "'!"); fwrite(text.data(), 1, text.size(), f); fclose(f); }
	}
};
int main(int argc, char** argv) { ConcreteTool tool(argc, argv); return tool.execute(); }
