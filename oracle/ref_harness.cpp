// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A driver of our own that links the reference's *own* hot-path translation units, compiled
// from where they lie under /root/reference/src (see oracle/Makefile, target `ref`):
//   bwt_search.cpp AlignmentCandidates.cpp nw_alignment.cpp tools.cpp KmerAnalysis.cpp
//   Mapping.cpp (only its per-read helpers survive --gc-sections) GetData.cpp bwt_index.cpp
//   BWT_Index/{bntseq,bwt,utils,...}.c
// The full reference binary is unbuildable in this image (its BAM writer needs the vendored
// htslib, which needs a generated config.h plus bzlib.h/lzma.h that are absent), so this file
// supplies what main.cpp (globals, flag parsing: main.cpp:9-18,96-205) and the body of
// ReadMapping (Mapping.cpp:579-681) supply in the reference, and nothing else: every
// algorithmic function called below is the reference's object code.
//
// One deliberate difference: ReadItem_t::sub_score/mis_num/mapq are zeroed
// before use. The reference leaves them uninitialised (UB); "sub_score = 0" is the contract.
//
// Modes:
//   ref_harness map  -i IDX -f R1 [-f2 R2] -o out.sam [-j junc.tab] [-mis N] [-max_dup N] [-p]
//                    [-m] [-unique] [-all_sj] [-max_intron N] [-min_intron N] [-dump stages.txt]
//   ref_harness nw            (stdin: "s1 s2" per line -> stdout "a1 a2")
//   ref_harness search -i IDX (stdin: read sequences, one per line -> per-start BWT_Search dump)
#include "structure.h"
#include <sys/stat.h>
#include <time.h>

// ---- globals that main.cpp defines in the reference (main.cpp:9-18) ----
bwt_t *Refbwt;
bwaidx_t *RefIdx;
char SJFileName[256];
unsigned int MaxDupNum;
const char* VersionStr = "1.4.6";
vector<string> ReadFileNameVec1, ReadFileNameVec2;
char *RefSequence, *IndexFileName, *OutputFileName;
bool bDebugMode, bSilent, bPairEnd, FastQFormat, bMultiHit, bUnique, bFindAllJunction, gzCompressed;
int iThreadNum, MaxInsertSize, MaxGaps, MaxIntronSize, MinIntronSize, MaxMismatch, OutputFileFormat;
const char* SpliceJunctionArr[4] = { "GT/AG", "CT/AC", "GC/AG", "CT/GC" };

// ---- per-read helpers defined (non-static) in Mapping.cpp ----
extern void SetSingleAlignmentFlag(ReadItem_t& read);
extern void SetPairedAlignmentFlag(ReadItem_t& read1, ReadItem_t& read2);
extern void EvaluateMAPQ(ReadItem_t& read);
extern void OutputPairedAlignments(ReadItem_t& read1, ReadItem_t& read2, int& myUniqueMapping, int& myUnMapping, int& myPairing, vector<string>& SamOutputVec);
extern void OutputSingledAlignments(ReadItem_t& read, int& myUniqueMapping, int& myUnMapping, vector<string>& SamOutputVec);
extern void RemoveRedundantCandidates(vector<AlignmentCandidate_t>& AlignmentVec);
extern bool CheckPairedAlignmentCandidates(vector<AlignmentCandidate_t>& AlignmentVec1, vector<AlignmentCandidate_t>& AlignmentVec2);
extern void RemoveUnMatedAlignmentCandidates(vector<AlignmentCandidate_t>& AlignmentVec1, vector<AlignmentCandidate_t>& AlignmentVec2);
extern void CheckPairedFinalAlignments(ReadItem_t& read1, ReadItem_t& read2);
extern void UpdateLocalSJMap(AlignmentCandidate_t& Aln, map<pair<int64_t, int64_t>, SpliceJunction_t>& LocalSJMap);
extern void UpdateGlobalSJMap(map<pair<int64_t, int64_t>, SpliceJunction_t>& LocalSJMap);
extern int OutputSpliceJunctions();
extern bool CheckReadFormat(const char* filename);
extern bwtint_t bwt_sa(bwtint_t k);

static FILE* dumpf = NULL;

static void dump_seeds(const char* tag, const char* hdr, vector<SeedPair_t>& v)
{
	fprintf(dumpf, "%s %s %d", tag, hdr, (int)v.size());
	for (size_t i = 0; i < v.size(); i++) fprintf(dumpf, " %d:%d:%lld", v[i].rPos, v[i].rLen, (long long)v[i].gPos);
	fprintf(dumpf, "\n");
}

static void dump_cands(const char* tag, const char* hdr, vector<AlignmentCandidate_t>& v)
{
	fprintf(dumpf, "%s %s %d", tag, hdr, (int)v.size());
	for (size_t i = 0; i < v.size(); i++) fprintf(dumpf, " %d:%lld:%d:%d", v[i].Score, (long long)v[i].PosDiff, v[i].PairedAlnCanIdx, (int)v[i].SeedVec.size());
	fprintf(dumpf, "\n");
}

static void dump_final(const char* hdr, int mate, ReadItem_t& r, vector<AlignmentCandidate_t>& v)
{
	fprintf(dumpf, "R%d %s score=%d sub=%d mapq=%d best=%d can=%d", mate, hdr, r.score, r.sub_score, r.mapq, r.iBestAlnCanIdx, r.CanNum);
	for (int i = 0; i < r.CanNum; i++)
	{
		fprintf(dumpf, " [%d,%d,%d", r.AlnReportArr[i].AlnScore, r.AlnReportArr[i].SJtype, r.AlnReportArr[i].PairedAlnCanIdx);
		if (r.AlnReportArr[i].AlnScore > 0) fprintf(dumpf, ",%d,%lld,%d,%s", r.AlnReportArr[i].coor.ChromosomeIdx, (long long)r.AlnReportArr[i].coor.gPos, r.AlnReportArr[i].coor.bDir ? 1 : 0, r.AlnReportArr[i].coor.CIGAR.c_str());
		fprintf(dumpf, "]");
	}
	fprintf(dumpf, "\n");
	if (r.score > 0 && r.iBestAlnCanIdx < (int)v.size())
	{
		vector<SeedPair_t>& s = v[r.iBestAlnCanIdx].SeedVec;
		fprintf(dumpf, "F%d %s %d", mate, hdr, (int)s.size());
		for (size_t i = 0; i < s.size(); i++) fprintf(dumpf, " %d:%d:%lld:%d:%d%d", s[i].rPos, s[i].rLen, (long long)s[i].gPos, s[i].gLen, s[i].bSimple ? 1 : 0, s[i].bAcceptorSite ? 1 : 0);
		fprintf(dumpf, "\n");
	}
}

static int run_map(int argc, char* argv[])
{
	int i;
	string parameter;

	MaxGaps = 5; MaxDupNum = 100; iThreadNum = 1; bPairEnd = false; bDebugMode = false; bMultiHit = false;
	bUnique = false; bSilent = true; bFindAllJunction = false; MaxIntronSize = 500000; MinIntronSize = 5;
	OutputFileName = (char*)"output.sam"; OutputFileFormat = 0; FastQFormat = true; MaxMismatch = 0;
	strcpy(SJFileName, "junctions.tab"); RefSequence = IndexFileName = NULL;
	const char* dumpname = NULL;

	for (i = 2; i < argc; i++)
	{
		parameter = argv[i];
		if (parameter == "-i") IndexFileName = argv[++i];
		else if (parameter == "-f") { while (++i < argc && argv[i][0] != '-') ReadFileNameVec1.push_back(argv[i]); i--; }
		else if (parameter == "-f2") { while (++i < argc && argv[i][0] != '-') ReadFileNameVec2.push_back(argv[i]); i--; }
		else if (parameter == "-t") ++i; // harness is single-threaded (canonical oracle configuration)
		else if (parameter == "-o") OutputFileName = argv[++i];
		else if (parameter == "-mis" && i + 1 < argc) MaxMismatch = atoi(argv[++i]);
		else if (parameter == "-max_dup" && i + 1 < argc)
		{
			MaxDupNum = (unsigned int)atoi(argv[++i]);
			if (MaxDupNum < 100) MaxDupNum = 100; else if (MaxDupNum >= 10000) MaxDupNum = 10000;
		}
		else if (parameter == "-silent") bSilent = true;
		else if (parameter == "-j") strcpy(SJFileName, argv[++i]);
		else if (parameter == "-p") bPairEnd = true;
		else if (parameter == "-m") bMultiHit = true;
		else if (parameter == "-unique") bUnique = true;
		else if (parameter == "-all_sj") bFindAllJunction = true;
		else if (parameter == "-max_intron") { if ((MaxIntronSize = atoi(argv[++i])) < 100000) MaxIntronSize = 100000; }
		else if (parameter == "-min_intron") MinIntronSize = atoi(argv[++i]);
		else if (parameter == "-dump") dumpname = argv[++i];
		else { fprintf(stderr, "Error! Unknow parameter: %s\n", argv[i]); return 1; }
	}
	if (ReadFileNameVec1.size() == 0 || IndexFileName == NULL) { fprintf(stderr, "need -i and -f\n"); return 1; }
	if (!CheckBWAIndexFiles(IndexFileName)) { fprintf(stderr, "Error! Please specify a valid reference index!\n"); return 1; }
	if (dumpname) dumpf = fopen(dumpname, "w");

	struct timespec ts_h0, ts_h1, ts_h2;                     // (harness only: wall clock of the index load and of the mapping loop, to stderr -- bench.py's `reference` CPU leg reads it)
	clock_gettime(CLOCK_MONOTONIC, &ts_h0);
	RefIdx = bwa_idx_load(IndexFileName);
	Refbwt = RefIdx->bwt;
	RestoreReferenceInfo();
	clock_gettime(CLOCK_MONOTONIC, &ts_h1);

	// ---- what Mapping() does around the threads (Mapping.cpp:738-751, 760-790) ----
	FILE* sam_out = fopen(OutputFileName, "w");
	fprintf(sam_out, "@PG\tID:Dart\tPN:Dart\tVN:%s\n", VersionStr);
	for (i = 0; i < (int)ChromosomeVec.size(); i++) fprintf(sam_out, "@SQ\tSN:%s\tLN:%lld\n", ChromosomeVec[i].name, (long long)ChromosomeVec[i].len);

	int64_t iTotalReadNum = 0, iUniqueMapping = 0, iUnMapping = 0, iPaired = 0;
	map<pair<int64_t, int64_t>, SpliceJunction_t> LocalSJMap;
	ReadItem_t* ReadArr = new ReadItem_t[ReadChunkSize];
	vector<string> SamOutputVec;
	vector<SeedPair_t> SeedPairVec1, SeedPairVec2;
	vector<AlignmentCandidate_t> AlignmentVec1, AlignmentVec2;

	for (int LibraryID = 0; LibraryID < (int)ReadFileNameVec1.size(); LibraryID++)
	{
		bool bSepLibrary;
		FILE *fh1 = NULL, *fh2 = NULL; gzFile gz1 = NULL, gz2 = NULL;
		string fn = ReadFileNameVec1[LibraryID];
		gzCompressed = (fn.substr(fn.find_last_of('.') + 1) == "gz");
		FastQFormat = CheckReadFormat(fn.c_str());
		if (gzCompressed) gz1 = gzopen(fn.c_str(), "rb"); else fh1 = fopen(fn.c_str(), "r");
		if (ReadFileNameVec1.size() == ReadFileNameVec2.size())
		{
			bSepLibrary = bPairEnd = true;
			if (FastQFormat != CheckReadFormat(ReadFileNameVec2[LibraryID].c_str())) { fprintf(stderr, "Error! different format\n"); return 1; }
			if (gzCompressed) gz2 = gzopen(ReadFileNameVec2[LibraryID].c_str(), "rb"); else fh2 = fopen(ReadFileNameVec2[LibraryID].c_str(), "r");
		}
		else bSepLibrary = false;
		if (fh1 == NULL && gz1 == NULL) continue;
		if (bSepLibrary && fh2 == NULL && gz2 == NULL) continue;

		// ---- the body of ReadMapping (Mapping.cpp:589-673), one thread ----
		while (true)
		{
			int ReadNum, j, myUniqueMapping, myUnMapping, myPairing;
			if (gzCompressed) ReadNum = gzGetNextChunk(bSepLibrary, gz1, gz2, ReadArr);
			else ReadNum = GetNextChunk(bSepLibrary, fh1, fh2, ReadArr);
			if (ReadNum == 0) break;
			for (i = 0; i < ReadNum; i++) { ReadArr[i].sub_score = 0; ReadArr[i].mis_num = 0; ReadArr[i].mapq = 0; ReadArr[i].AlnReportArr = NULL; } // F6 contract
			if (bPairEnd && ReadNum % 2 == 0)
			{
				for (i = 0, j = 1; i != ReadNum; i += 2, j += 2)
				{
					SeedPairVec1 = IdentifySeedPairs(ReadArr[i].rlen, ReadArr[i].EncodeSeq);
					if (dumpf) dump_seeds("S1", ReadArr[i].header, SeedPairVec1);
					AlignmentVec1 = GenerateAlignmentCandidate(ReadArr[i].rlen, SeedPairVec1);
					SeedPairVec2 = IdentifySeedPairs(ReadArr[j].rlen, ReadArr[j].EncodeSeq);
					if (dumpf) dump_seeds("S2", ReadArr[j].header, SeedPairVec2);
					AlignmentVec2 = GenerateAlignmentCandidate(ReadArr[j].rlen, SeedPairVec2);
					if (CheckPairedAlignmentCandidates(AlignmentVec1, AlignmentVec2)) RemoveUnMatedAlignmentCandidates(AlignmentVec1, AlignmentVec2);
					RemoveRedundantCandidates(AlignmentVec1); RemoveRedundantCandidates(AlignmentVec2);
					if (dumpf) { dump_cands("C1", ReadArr[i].header, AlignmentVec1); dump_cands("C2", ReadArr[j].header, AlignmentVec2); }
					GenMappingReport(true, ReadArr[i], AlignmentVec1);
					GenMappingReport(false, ReadArr[j], AlignmentVec2);
					CheckPairedFinalAlignments(ReadArr[i], ReadArr[j]);
					SetPairedAlignmentFlag(ReadArr[i], ReadArr[j]);
					EvaluateMAPQ(ReadArr[i]); EvaluateMAPQ(ReadArr[j]);
					if (dumpf) { dump_final(ReadArr[i].header, 1, ReadArr[i], AlignmentVec1); dump_final(ReadArr[j].header, 2, ReadArr[j], AlignmentVec2); }
					if (ReadArr[i].mapq == 50 || (bFindAllJunction && ReadArr[i].score > 0)) UpdateLocalSJMap(AlignmentVec1[ReadArr[i].iBestAlnCanIdx], LocalSJMap);
					if (ReadArr[j].mapq == 50 || (bFindAllJunction && ReadArr[j].score > 0)) UpdateLocalSJMap(AlignmentVec2[ReadArr[j].iBestAlnCanIdx], LocalSJMap);
				}
			}
			else
			{
				for (i = 0; i != ReadNum; i++)
				{
					SeedPairVec1 = IdentifySeedPairs(ReadArr[i].rlen, ReadArr[i].EncodeSeq);
					if (dumpf) dump_seeds("S1", ReadArr[i].header, SeedPairVec1);
					AlignmentVec1 = GenerateAlignmentCandidate(ReadArr[i].rlen, SeedPairVec1);
					RemoveRedundantCandidates(AlignmentVec1);
					if (dumpf) dump_cands("C1", ReadArr[i].header, AlignmentVec1);
					GenMappingReport(true, ReadArr[i], AlignmentVec1);
					SetSingleAlignmentFlag(ReadArr[i]); EvaluateMAPQ(ReadArr[i]);
					if (dumpf) dump_final(ReadArr[i].header, 1, ReadArr[i], AlignmentVec1);
					if (ReadArr[i].mapq == 50 || (bFindAllJunction && ReadArr[i].score > 0)) UpdateLocalSJMap(AlignmentVec1[ReadArr[i].iBestAlnCanIdx], LocalSJMap);
				}
			}
			myUniqueMapping = myUnMapping = myPairing = 0; SamOutputVec.clear();
			if (bPairEnd && ReadNum % 2 == 0) for (i = 0; i != ReadNum; i += 2) OutputPairedAlignments(ReadArr[i], ReadArr[i + 1], myUniqueMapping, myUnMapping, myPairing, SamOutputVec);
			else for (i = 0; i != ReadNum; i++) OutputSingledAlignments(ReadArr[i], myUniqueMapping, myUnMapping, SamOutputVec);
			iTotalReadNum += ReadNum; iUniqueMapping += myUniqueMapping; iUnMapping += myUnMapping; iPaired += myPairing;
			for (vector<string>::iterator iter = SamOutputVec.begin(); iter != SamOutputVec.end(); iter++) fprintf(sam_out, "%s\n", iter->c_str());
			for (i = 0; i != ReadNum; i++)
			{
				delete[] ReadArr[i].header; delete[] ReadArr[i].seq; delete[] ReadArr[i].EncodeSeq;
				if (FastQFormat) delete[] ReadArr[i].qual;
				delete[] ReadArr[i].AlnReportArr;
			}
		}
		if (fh1) fclose(fh1); if (fh2) fclose(fh2); if (gz1) gzclose(gz1); if (gz2) gzclose(gz2);
	}
	UpdateGlobalSJMap(LocalSJMap);
	fclose(sam_out);
	clock_gettime(CLOCK_MONOTONIC, &ts_h2);
	fprintf(stderr, "[ref_harness] index load %.3f s, mapping phase %.3f s, %lld reads, 1 thread\n", (ts_h1.tv_sec - ts_h0.tv_sec) + 1e-9 * (ts_h1.tv_nsec - ts_h0.tv_nsec),
	        (ts_h2.tv_sec - ts_h1.tv_sec) + 1e-9 * (ts_h2.tv_nsec - ts_h1.tv_nsec), (long long)iTotalReadNum);
	if (dumpf) fclose(dumpf);
	// stats text, Mapping.cpp:812-822
	if (iTotalReadNum > 0)
	{
		if (bPairEnd) fprintf(stdout, "\t# of total mapped reads = %lld (sensitivity = %.2f%%)\n\t# of paired sequences = %lld (%.2f%%)\n", (long long)(iTotalReadNum - iUnMapping), (int)(10000 * (1.0*(iTotalReadNum - iUnMapping) / iTotalReadNum) + 0.5) / 100.0, (long long)iPaired, (int)(10000 * (1.0*iPaired / iTotalReadNum) + 0.5) / 100.0);
		else fprintf(stdout, "\t# of total mapped reads = %lld (sensitivity = %.2f%%)\n", (long long)(iTotalReadNum - iUnMapping), (int)(10000 * (1.0*(iTotalReadNum - iUnMapping) / iTotalReadNum) + 0.5) / 100.0);
		fprintf(stdout, "\t# of unique mapped reads = %lld (%.2f%%)\n", (long long)iUniqueMapping, (int)(10000 * (1.0*iUniqueMapping / iTotalReadNum) + 0.5) / 100.0);
		if (!bUnique) fprintf(stdout, "\t# of multiple mapped reads = %lld (%.2f%%)\n", (long long)(iTotalReadNum - iUnMapping - iUniqueMapping), (int)(10000 * (1.0*(iTotalReadNum - iUnMapping - iUniqueMapping) / iTotalReadNum) + 0.5) / 100.0);
		fprintf(stdout, "\t# of unmapped reads = %lld (%.2f%%)\n", (long long)iUnMapping, (int)(10000 * (1.0*iUnMapping / iTotalReadNum) + 0.5) / 100.0);
		i = OutputSpliceJunctions();
		fprintf(stdout, "\t# of splice junctions = %d (file: %s)\n", i, SJFileName);
		fprintf(stdout, "\tAlignment output: %s\n\n", OutputFileName);
	}
	return 0;
}

static int run_nw()
{
	char a[4096], b[4096];
	while (scanf("%4095s %4095s", a, b) == 2)
	{
		string s1 = a, s2 = b;
		nw_alignment((int)s1.length(), s1, (int)s2.length(), s2);
		printf("%s %s\n", s1.c_str(), s2.c_str());
	}
	return 0;
}

static int run_search(int argc, char* argv[])
{
	MaxDupNum = 100; iThreadNum = 1;
	for (int i = 2; i < argc; i++)
	{
		if (!strcmp(argv[i], "-i")) IndexFileName = argv[++i];
		else if (!strcmp(argv[i], "-max_dup")) MaxDupNum = atoi(argv[++i]);
	}
	RefIdx = bwa_idx_load(IndexFileName); Refbwt = RefIdx->bwt;
	char line[8192];
	while (fgets(line, sizeof line, stdin))
	{
		int rlen = (int)strlen(line); while (rlen > 0 && (line[rlen - 1] == '\n' || line[rlen - 1] == '\r')) rlen--;
		if (rlen == 0) continue;
		uint8_t* enc = new uint8_t[rlen];
		for (int i = 0; i < rlen; i++) enc[i] = nst_nt4_table[(int)(unsigned char)line[i]];
		// every start position, as BWT_Search(seq,start,rlen) is called by IdentifySeedPairs
		for (int start = 0; start < rlen; start++)
		{
			if (enc[start] > 3) continue;
			bwtSearchResult_t r = BWT_Search(enc, start, rlen);
			printf("%d:%d:%d", start, r.freq > 0 ? r.len : -1, r.freq);
			for (int j = 0; j < r.freq; j++) printf(",%llu", (unsigned long long)r.LocArr[j]);
			printf(" ");
			if (r.LocArr) delete[] r.LocArr;
		}
		printf("\n");
		delete[] enc;
	}
	return 0;
}

int main(int argc, char* argv[])
{
	if (argc >= 2 && !strcmp(argv[1], "map")) return run_map(argc, argv);
	if (argc >= 2 && !strcmp(argv[1], "nw")) return run_nw();
	if (argc >= 2 && !strcmp(argv[1], "search")) return run_search(argc, argv);
	fprintf(stderr, "usage: ref_harness map|nw|search ...\n");
	return 2;
}
