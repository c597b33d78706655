// MappingQC — drop-in for src/MappingQC/main.cpp:21-188 on the MI355X path: same flags, defaults, checks and output
// (qcML / TXT), including the contamination check, -somatic_custom_bed and -read_qc.
#include "Statistics.hpp"
using namespace ngsbits;

class ConcreteTool : public ToolBase
{
public:
	ConcreteTool(int argc, char** argv) : ToolBase(argc, argv) {}
	void setup() override
	{
		setDescription("Calculates QC metrics based on mapped NGS reads.");
		addInfile("in", "Input BAM/CRAM file.", false, true);
		addOutfile("out", "Output qcML file. If unset, writes to STDOUT.", true);
		addInfile("roi", "Input target region BED file (for panel, WES, etc.).", true, true);
		addFlag("wgs", "WGS mode without target region. Genome information is taken from the BAM/CRAM file.");
		addFlag("rna", "RNA mode without target region. Genome information is taken from the BAM/CRAM file.");
		addFlag("txt", "Writes TXT format instead of qcML.");
		addInt("min_mapq", "Minmum mapping quality to consider a read mapped.", true, 1);
		addFlag("no_cont", "Disables sample contamination calculation, e.g. for tumor or non-human samples.");
		addFlag("debug", "Enables verbose debug outout.");
		addEnum("build", "Genome build used to generate the input (needed for WGS and contamination only).", true, {"hg19", "hg38", "non_human"}, "hg38");
		addInfile("ref", "Reference genome FASTA file. If unset 'reference_genome' from the 'settings.ini' file is used.", true, false);
		addFlag("cfdna", "Add additional QC parameters for cfDNA samples. Only supported mit '-roi'.");
		addInfile("somatic_custom_bed", "Somatic custom region of interest (subpanel of actual roi). If specified, additional depth metrics will be calculated.", true, true);
		addOutfile("read_qc", "If set, a read QC file in qcML format is created (just like ReadQC/SeqPurge).", true);
		addFlag("single_end", "Enable single-end mode. Use for ONT, PacBio and Roche. Illumina single-end data is auto-detected based on paired reads.");
		addFlag("no_ref", "[ngsqc extension] Run without a reference genome: GC/AT dropout become n/a, no N-base correction in WGS mode.");
		// --changelog (src/MappingQC/main.cpp)
		changeLog(2026, 7, 7, "Added support for short-read single-end (Roche). Renamed long_read parameter to single_end.");
		changeLog(2023, 11, 8, "Added long_read support.");
		changeLog(2023, 5, 12, "Added 'read_qc' parameter.");
		changeLog(2022, 5, 25, "Added new QC metrics to WGS mode.");
		changeLog(2021, 2, 9, "Added new QC metrics for uniformity of coverage (QC:2000057-QC:2000061).");
		changeLog(2020, 11, 27, "Added CRAM support.");
		changeLog(2018, 7, 11, "Added build switch for hg38 support.");
		changeLog(2018, 3, 29, "Removed '3exons' flag.");
		changeLog(2016, 12, 20, "Added support for spliced RNA reads (relevant e.g. for insert size)");
	}
	void main() override
	{
		std::string roi_file = getInfile("roi");
		bool wgs = getFlag("wgs"), rna = getFlag("rna");
		std::string in = getInfile("in");
		std::string ref_file = getInfile("ref");
		if (getFlag("no_ref")) ref_file = NO_REF;
		if (ref_file == "") ref_file = settingsString("reference_genome");
		if (ref_file == "") NB_THROW(CommandLineParsingException, "Reference genome FASTA unset in both command-line and settings.ini file!");
		bool cfdna = getFlag("cfdna"); int min_mapq = getInt("min_mapq");
		int parameters_set = (roi_file != "" ? 1 : 0) + wgs + rna;
		if (parameters_set != 1) NB_THROW(CommandLineParsingException, "You have to use exactly one of the parameters 'roi', 'wgs', or 'rna' !");
		if (cfdna && roi_file == "") NB_THROW(CommandLineParsingException, "The flag 'cfdna' can only be used with parameter 'roi'!");
		// The reference reads the BAM once per pass (read QC main.cpp:80-98, mapping :100-141, contamination :143-151, somatic sub-panel
		// :153-165). Here the passes behind the mapping pass are announced first, so that the mapping call runs ONE fused GPU job over
		// the BAM and the others take their counts from it (Statistics::planFused); the outputs are the same.
		std::string read_qc = trimmed(getOutfile("read_qc"));
		std::string somatic_custom_roi_file = getInfile("somatic_custom_bed");
		const bool do_cont = !getFlag("no_cont") && getEnum("build") != "non_human";
		FusedPlan plan;
		plan.contamination = do_cont; plan.build = getEnum("build"); plan.roi_file = roi_file; plan.include_not_properly_paired = getFlag("single_end");
		plan.read_qc = !read_qc.empty(); plan.single_end = getFlag("single_end");
		if (somatic_custom_roi_file != "") { plan.somatic = true; plan.somatic_bed.load(somatic_custom_roi_file); plan.somatic_bed.merge(); plan.somatic_min_mapq = min_mapq; }
		Statistics::planFused(in, plan);

		std::vector<std::string> parameters; QCCollection metrics;
		if (wgs)
		{
			std::string build = getEnum("build");
			if (build == "non_human") metrics = Statistics::mapping(in, ref_file, min_mapq);
			else metrics = Statistics::mapping_wgs(in, resourceDir() + "/" + (build == "hg19" ? "hg19_439_omim_genes.bed" : "hg38_440_omim_genes.bed"), min_mapq, ref_file);
			parameters.push_back("-wgs");
		}
		else if (rna) { metrics = Statistics::mapping(in, ref_file, min_mapq); parameters.push_back("-rna"); }
		else
		{
			BedFile roi; roi.load(roi_file); roi.merge();
			metrics = Statistics::mapping(roi, in, ref_file, min_mapq, cfdna);
			parameters.push_back("-roi"); parameters.push_back(fileName(roi_file));
			if (cfdna) parameters.push_back("-cfdna");
		}
		// raw read QC (main.cpp:80-98)
		if (!read_qc.empty())
		{
			StatisticsReads stats(getFlag("single_end"));
			if (!stats.takeFused(in)) { BamReader reader(in, ref_file); stats.update(reader); }
			stats.getResult().storeToQCML(read_qc, {in}, "", "MappingQC", version());
		}
		// sample contamination (main.cpp:143-151)
		QCCollection metrics_cont;
		if (do_cont) metrics_cont = Statistics::contamination(getEnum("build"), in, ref_file, roi_file, getFlag("debug"), 20, 50, getFlag("single_end"));
		// somatic sub-panel depth (main.cpp:153-165)
		if (somatic_custom_roi_file != "")
		{
			metrics.insert(Statistics::somaticCustomDepth(plan.somatic_bed, in, ref_file, min_mapq));
			parameters.push_back("-somatic_custom_bed " + somatic_custom_roi_file);
		}
		Statistics::clearFused();
		if (getFlag("single_end")) parameters.push_back("-single_end");
		std::string out = getOutfile("out");
		if (getFlag("txt"))
		{
			std::vector<std::string> output; metrics.appendToStringList(output); output.push_back(""); metrics_cont.appendToStringList(output);
			std::string text; for (auto& l : output) text += l + "\n";
			if (out.empty()) fwrite(text.data(), 1, text.size(), stdout);
			else { FILE* f = fopen(out.c_str(), "wb"); if (!f) NB_THROW(FileAccessException, "Could not open file for writing: '" + out + "'!"); fwrite(text.data(), 1, text.size(), f); fclose(f); }
		}
		else
		{
			metrics.insert(metrics_cont);
			metrics.storeToQCML(out, {in}, join(parameters, " "), "MappingQC", version());
		}
	}
};
int main(int argc, char** argv) { ConcreteTool tool(argc, argv); return tool.execute(); }
