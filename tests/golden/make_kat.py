#!/usr/bin/env python3
"""Writes tests/golden/kat_reference.json: the known-answer vectors the reference's own JUnit tests hold for
the seed-and-extend path (inputs and expected outputs transcribed as data; T/ = /root/reference/src/test/java/).

Each case cites the test it comes from.  The JSON is what the tests read; this script only documents how the
composite strings (prefix + shared + ..., reverse complements) of the JUnit sources were assembled.
"""
import json
import os

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "R": "Y", "Y": "R", "S": "S", "W": "W", "K": "M", "M": "K"}


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def P(**kw):
    return kw


# T/AlignerWorker_Test.java:788-799 makeParameters()
AW = P(MutationPenalty=1, InsertionStart_Penalty=1.5, InsertionExtension_Penalty=0.6, DeletionStart_Penalty=1.5, DeletionExtension_Penalty=0.5,
       MaxErrorRate=0.2, AmbiguityPenalty=0.2, UnalignedPenalty=0.2, Max_PenaltySpan=0)
# the six "rounding error" cases :241-481
RE = P(MutationPenalty=6, InsertionStart_Penalty=9, InsertionExtension_Penalty=5, DeletionStart_Penalty=6, DeletionExtension_Penalty=5,
       MaxErrorRate=1, AmbiguityPenalty=1, UnalignedPenalty=1, Max_PenaltySpan=0)
LOW = P(MutationPenalty=1, InsertionStart_Penalty=1.5, InsertionExtension_Penalty=0.6, DeletionStart_Penalty=1.5, DeletionExtension_Penalty=0.5,
        MaxErrorRate=0.05, AmbiguityPenalty=0.05, UnalignedPenalty=0.05, Max_PenaltySpan=0)

align_cases = []


def case(name, cite, ref, mates, params, expected_inner=0.0, deviation=1.0, **expect):
    align_cases.append(dict(name=name, cite=cite, reference=ref, mates=mates, params=params, expectedInner=expected_inner,
                            deviation=deviation, expect=expect))


# ---- T/AlignerWorker_Test.java (Api.alignOnce)
case("testIndelNotDuplicated", "T/AlignerWorker_Test.java:10-16", "TTAAACAGATCACCTCGCTGAGCGGGT", ["TTAAACAGATCACCCGCTGAGCGGGT"], AW, num=1)
case("testPartialAmbiguity", "T/AlignerWorker_Test.java:18-31", "AACAGGCGGT" + "AACARGCGGT" + "AACARRCGGT", ["AACAAGCGGT"], AW, num=1,
     alignedB0="AACARGCGGT")
_identical = "GGGGTCAC"
_q = _identical + "AAAA"
case("testHashblockAlsoMatchingNearEndOfContig", "T/AlignerWorker_Test.java:40-49", _identical + "CAAA" + "TCTCGGAGAGCTCGA" + _q + "T", [_q], AW,
     num=1, alignedB0=_q)
case("testFirstHashblockMultipleGoodMatches", "T/AlignerWorker_Test.java:51-61", "AACGATTTGG" + "AACGATCGCG" + "G", ["AACGATCGGG"], AW, num=1,
     alignedB0="AACGATCGCG")

q1p, q1m, ov, ovm, q2s = "AACGAGTG", "AAGGACAG", "AACGACGGTT", "AACGAGCGTT", "AAAGACCC"
case("testOverlappingPairedEndQueriesFewerMutationsOverlappingBothQueries", "T/AlignerWorker_Test.java:63-99",
     (q1m + ov + q2s) + (q1p + ovm + q2s), [q1p + ov, rc(ov + q2s)], AW, 0, 1000000, num=1, alignedB0=q1p + ovm)

q1t = "ACGTGAACCGGTTAAACCC"
sep = "ACAGTTGGCGAGCGC"
case("testOverlappingPairedEndQueriesBetterThanSurprisingOffset", "T/AlignerWorker_Test.java:101-146", q1t + sep + q1t + "C", [q1t, rc(q1t)], AW,
     0, len(sep) // 2, num=2, startsB=[[0, 0], [34, 34]])

prefix, shared, sharedM, suffix = "ACGTACGTCC", "AACCGGTTGG", "AACCTGTTGG", "AAACCCGGGTTT"
cm = prefix + sharedM + suffix
case("testOverlappingPairedEndQueriesMultipleMatches", "T/AlignerWorker_Test.java:148-175", "GGGG" + cm + cm + "TTTT",
     [prefix + shared, rc(shared + suffix)], AW, 0, float(len(cm)), num=2)

shared = "AACCGGTTCACTCGGGACACACACC" + "ACGTCGTATTGTGCGCCGTTACAAA" + "GTTTGTTTAGAGCCCCTTTTAGCGA"
sharedM = "AACTGGTTCACTCGGGACACACACC" + "ACGTCGTAATGTGCGCCGTTACAAA" + "GTTTGTTTAGAGCCCCTCTTAGCGA"
cm = sharedM
case("testMultipleCandidateMatches", "T/AlignerWorker_Test.java:177-204", "GGGG" + cm + "AAAA" + cm + "TTTT", [shared, rc(shared)], AW,
     -1.0 * len(cm), float(len(cm) // 4), num=2)

shared = "GACATTGGCAAAGTCAACAAAGCGGAAATCAAGGAAGCCATGGACGGCGTATTGAAGAAGATGCAGGGCTTTGACTTTACCAAATTCAAGGAAGAACTTGGTAAGAGAGGTTTTAAAGTCCGGGAAGCCAGGGCAAGCACCGGGAAACTC"
cm = "T" + shared
case("testMultipleCandidateMatches2", "T/AlignerWorker_Test.java:206-239", "C" + cm + "" + cm + "TTTT", ["G" + shared, rc(shared)],
     P(MutationPenalty=6, InsertionStart_Penalty=9, InsertionExtension_Penalty=5.4, DeletionStart_Penalty=9, DeletionExtension_Penalty=4.5,
       MaxErrorRate=1.2, AmbiguityPenalty=1.2, UnalignedPenalty=1.2, Max_PenaltySpan=0), -1.0 * len(cm), float(len(cm) // 4 // 6), num=2)


def rounding(name, cite, q1, q2fwd, cm):
    case(name, cite, "ACGT" + cm + cm + "ACGT", [q1, rc(q2fwd)], RE, -1.0 * len(cm), float(len(cm) // 4 // 6), num=2)


prefix = "AAACCCGGGTTTAAAACCCCGGGGTTTTAAAAACCCCCGGGGG"
shared = "GACATTGGCAAAGTCAACAAAGCGGAAATCAAGGAAGCCATGGACGGGGTATTGAAGAAGATGCAGGGCTTTGACTTTACCAAATTCAAGGAAGAACTTGGTAAGAG"
sharedM = "GACATTGGCAAAGTCAACAAAGCGGAAATCAAGGAAGCCATGGACGGCGTATTGAAGAAGATGCAGGGCTTTGACTTTACCAAATTCAAGGAAGAACTTGGTAAGAG"
suffix = "AGGTTTTAAAGTCCGGGAAGCCAGGGCAAGCACCGGGAAACTC"
rounding("testPairedEndQueriesRoundingError", "T/AlignerWorker_Test.java:241-277", prefix + sharedM, shared + suffix, prefix + shared + suffix)

prefix = "ATCCTTGATTTTCCCTTTAAGGGCGTTTATAATCCACCCTTTCGGATTGTTCTTTTCTCGTGATTTTCCGTTTAGGAGAGCCAGTTCTCCGATAAGGTCGGTTATCTTTTCTTGTGCCGTTATGAATGTCTCTTTGTTCCGGTTTAT"
shared = "CTC"
suffix = "TTCCGATGTGAAGCCGCAGGAATAACGGAGGTACTCGTACACATGGCTGTCTATCTGATATCGTGCTGTAACCTTTGCTTGCAATTCTTTCCCTTCCAGTTCTTCATCTCTGAACTGTGGGTGATAGACCGGGTAGAACCTAAACC"
suffixM = "TTCCGATGTGAAGCCGCAGGAATAACGGAGGTACTCGTACACATGGCTGTCTATATGATATCGTGCTGTAACCTTTGCTTGCAATTCTTTCCCTTCCAGTTCTTCATCTCTGAACTGTGGGTGATAGACCGGGTAGAACCTAAACC"
rounding("testPairedEndQueriesRoundingError2", "T/AlignerWorker_Test.java:279-316", prefix + shared, shared + suffixM, prefix + shared + suffix)

prefix = "GAACTGGAAGGGAAAGAAT"
shared = "TGCAAGCAAAGGTTACAGCACGATATCAGATAGACAGCCATGTGTACGAGTACCTCCGTTATTCCTGCGGCTTCACATCGGAAGAGATAAACCGGAACAAAGAGACATTCATAACGGAACAAGAAAAGATA"
sharedM = "TGCAAGCAAAGGTTACAGCACGATATCAGATAGACAGCCATGTGTACGAGTACCTCCGTTATTCCTGCGGCTTCACATCGGAAGAGATAAACCGGAACAAAGAGACATTCATAACGGCACAAGAAAAGATA"
suffix = "ACCGACCTTATCGGAGA"
rounding("testPairedEndQueriesRoundingError3", "T/AlignerWorker_Test.java:318-356", prefix + sharedM, shared + suffix, prefix + shared + suffix)

prefix = "GAACAAGGCACATGACGGTCTGGAAAACAATCCGGGAAAAGACGGCAAACT"
prefixM = "GAACAAGGCACATGACGGTCTGGAAAACAATCCAGGAAAAGACGGCAAACT"
shared = "GTTTTCAGACAAACACCCCTACATTACTGAAGCGCATCCGGGAGCAAAAAAAGCCGTGGACGCACTGACCAGGCGCATCAACGAAATGATAGCCGAAAT"
suffix = "GCCGGACAACCTGACGCTGGAGGAAAAAACCGACATCGCCCGCAACAATCT"
suffixM = "GTCGGACAACCTGACGCTGGAGGAAAAAACCGACATCGCCCGCAACAATCT"
rounding("testPairedEndQueriesRoundingError4", "T/AlignerWorker_Test.java:358-398", prefixM + shared, shared + suffixM, prefix + shared + suffix)

prefix = "TCTTTGTAGGGTGAAAGAGAAACCCATAAACGGGGATAGATTGAATGCTGGGAAGCATAAACAATC"
shared = "GGGGTAAGGTTAGCGAACCTTGCCTTTCATCCCCCATTATAACTTTACATAGAGGAACTTTATCTATCCCCCCCCGCCCCCAAA"
sharedM = "GGGGTAAGGTTAGCGTACCTTGCCTTTGATCCCCCATTATAACTTTACATAGAGGAACTTTATCTATCCCCCCCCGCCCCCAAA"
suffix = "GGGGGAGCGACCAAACGGCAGCTTCACTCAATGGAGTGTTACAGTTCATCAAAACCAAGTGATAAC"
rounding("testPairedEndQueriesRoundingError5", "T/AlignerWorker_Test.java:400-438", prefix + shared, sharedM + suffix, prefix + shared + suffix)

prefix = "CAATAGGGAGATAACAGCACAAAGGATTGAGTAGAACGAAATTCGTTTGTCCACATAACCGCCGTTTTTCAT"
suffixM = "TGTACCTTTCGGGCTGTTGCGTCCTCTATGCGCTTCGTATAGACTTCAACACGCTTTAGTTCTTGATACACC"
suffix = "TGTACCTTTCGGGCTGTTGCGTCCTCTATGCGCTTCGTATAGACTTCAACACGCTTTAGTTCTTGATACACC"
sharedM = "TCTGTACCCCTGCCGTTCAAAGTCCGCCAACACGTTTTTAGGCGATTTTCGGCACTTTCTAGGCTTTTCCCGTCTATT"
shared = "TCTGTACCCCTGCCGTTCAAAGTCCGCCAACACGTTTTTTAGGCGATTTTCGGCACTTTCAAGGCTTTTCCCGTCTATT"
rounding("testPairedEndQueriesRoundingError6", "T/AlignerWorker_Test.java:440-481", prefix + sharedM, sharedM + suffixM, prefix + shared + suffix)

shared = "CTTCCATATCTGTTTGCTTTTAAATTCAGCACAAAGATAGCTATATTTCAATAAAATACAAACATTTTGTACACAAACGTGTACACGCCATAAAAACCCGTTTCCAATCCTACCGCCCGTTGGTTGGTTTTGCTTTGCTCTTTTTCCC"
sharedM = "ATGCTTCCATATCTGTTTGCTTTTAAATTCAGCACAAAGATAGCTATATTTCAATAAAATACAAACATTTTGTACACAAACGTGTACACGCCATAAAAACCCGTTTCCAATCCTACCGCCCGTTGGTTGGTTTTGCTTTGCTCTTTTTCCCT"
cm = sharedM
case("testPairedEndQueriesOverlappingIndel", "T/AlignerWorker_Test.java:483-520", "ACGT" + cm + "AACCGGTT" + cm + "ACGT",
     [shared + "CT", rc("AG" + shared)],
     P(MutationPenalty=6, InsertionStart_Penalty=3, InsertionExtension_Penalty=2, DeletionStart_Penalty=3, DeletionExtension_Penalty=2, MaxErrorRate=1,
       AmbiguityPenalty=1, UnalignedPenalty=1, Max_PenaltySpan=0), -1.0 * len(cm), float(len(cm) // 4 // 6), num=2)

prefix = "TCTCGGCTGGCGGCAAGAGAAGAGAACACCTCGTGCAT"
shared = "AGGCTCGCCGTTCTCTAACCAGTAAACACAATATTCGACCATAACAGTTTTATCATTTATCGTTGTAATGCCCCTCTACCTCCAAGATGTAGACCTCTACCACTTCCTCGTA"
sharedM = "AGGCTCGCCGTTCTCTAACCAGTAAACACAATATTCGACCATAACAGTTTTATCATTTATCGTTGTAATGCCCCCTCTACCTCCAAGATGTAGACCTCTACCACTTCCTCGTA"
suffix = "AATGTCATAGATTATCCGGTCATGGGCGGTAATGTGT"
cm = prefix + shared + suffix
case("testPairedEndQueriesOverlappingInsertion", "T/AlignerWorker_Test.java:522-561", "ACGT" + cm + "ACGT" + cm + "ACGT",
     [prefix + sharedM, rc(sharedM + suffix)], LOW, -1.0 * len(shared), 0.5, num=2)

prefix, prefixM = "AACCGGTT", "AACCGG"
shared = "GACATTGGCAAAGTCAACAAAGCGGAAATCAAGGAAGCCATGGACGGCGTATTGAAGAAGATGCAGGGCTTTGACTTTACCAAATTCAAGGAAGAACTTGGTAAGAGAGGTTTTAAAGTCCGGGAAGCCAGGGCAAGCACCGGGAAACTC"
suffix, suffixM = "AACCGGTT", "CCGGTT"
cm = prefixM + shared + suffixM
case("testPairedEndQueriesWithIndelsNextToOverlap", "T/AlignerWorker_Test.java:563-600", "ACGT" + cm + "ACGT" + cm + "ACGT",
     [prefix + shared, rc(shared + suffix)], LOW, -1.0 * len(cm), 1.0, num=2)

prefix = "ACCGTAACAACCTCGCAGCGTCTTTCACCAAAGCTGACAATGGCGAGCAGGTACTAATTCGCA"
suffix = "GAAAAACGAGATTTACGCTTTGGTAAAAGTTGGTCGTGAAGATTTGATGATAACCCCGGAGCTGCAAGCAAGGATTGACAAGGCAAG"
m = prefix + "G" + suffix
case("testDeletionInMiddleOfQueryWithMultipleAlignments", "T/AlignerWorker_Test.java:602-624", "A" + m + m + "A", [prefix + suffix], AW, num=2)

case("queryExtendingPastEndOfReference", "T/AlignerWorker_Test.java:626-642",
     "GACCGGATATTCTGGTAATGACCCTTCAATTATAGACGTGAATGGTATCCAGCCGGGAGTAGATAGTAATAGTGCTTATCCTACAGCAACTCAATTGAGTTTAGGTGTGAC",
     ["ATCCTACAGCAACTCAATTGAGTTTAGGTGTGACTCTTCGCTTCAAATAAATGAGAAACAAATTATTAAAAATATGAAAGATATGAAATATATAAAATGTC"], AW, num=1,
     alignedB0="ATCCTACAGCAACTCAATTGAGTTTAGGTGTGAC")

case("testCustomParameters", "T/AlignerWorker_Test.java:644-672", "CGCGTACTCT", ["ACGCATCCTCTTTT"],
     P(MutationPenalty=1, InsertionStart_Penalty=0.8, InsertionExtension_Penalty=1, DeletionStart_Penalty=0.8, DeletionExtension_Penalty=1,
       MaxErrorRate=0.7, AmbiguityPenalty=0.9, UnalignedPenalty=0.9, Max_PenaltySpan=0), num=1, alignedB0="CGCGTACTCT")

refPrefix = "A" * 77
qPrefix, qPrefixM = "AACACACGGTGTTCAC", "AACCCACGGTGTTCAC"
insertion = "CACCCGCCCGCGCGCTCTCTCG"
sharedSuffix = "AATAACCGCCGGCGGTTATTAAAACCCCGGGGTTTTAAACCCGGGTTTAACCGGTTACGT"
refSuffix = "A" * 87
lp = dict(AW)
lp.update(InsertionExtension_Penalty=0.2, DeletionExtension_Penalty=0.2, MutationPenalty=2)
case("testLongCheapIndel", "T/AlignerWorker_Test.java:674-695", refPrefix + qPrefixM + sharedSuffix + qPrefix + refSuffix,
     [qPrefix + insertion + sharedSuffix], lp, num=1, alignedB0=qPrefixM + "-" * len(insertion) + sharedSuffix)

sp = dict(AW)
sp.update(Max_PenaltySpan=1)
shared = "AACCACAC"
case("test_maxPenaltySpan_with_perfectAlignment", "T/AlignerWorker_Test.java:697-710", shared + "AAAA" + shared + "AAGA", [shared + "AAAA"], sp, num=2)

# doTestPairedEndQueries :712-744
ref = "AAAAAAAAAAACGGAAAGAAATAACTTAAACGAACTAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAACGGAAAGAAATAAA"
s1, s2 = "CGGAAAGAAA", "CTTAAACGAACT"
case("testPairedEndQueries/query1", "T/AlignerWorker_Test.java:712-727", ref, [s1], AW, num=2)
case("testPairedEndQueries/query2rev", "T/AlignerWorker_Test.java:728-731", ref, [rc(s2)], AW, num=1)
case("testPairedEndQueries/query2fwd", "T/AlignerWorker_Test.java:728-731", ref, [s2], AW, num=1)
case("testPairedEndQueries/combined-reversed", "T/AlignerWorker_Test.java:733-743", ref, [s1, rc(s2)], AW, 3, 1.0, num=1)
case("testPairedEndQueries/combined-forward", "T/AlignerWorker_Test.java:733-743", ref, [s1, s2], AW, 3, 1.0, num=0)

# ---- T/PathAligner_Test.java:10-39,76-87 and T/HashBlockAligner_Test.java:10-48,84-96
PA = P(MutationPenalty=1, InsertionStart_Penalty=2, InsertionExtension_Penalty=0.5, DeletionStart_Penalty=2, DeletionExtension_Penalty=0.5,
       MaxErrorRate=1, AmbiguityPenalty=0.1, UnalignedPenalty=0.1, Max_PenaltySpan=0)
PA3 = dict(PA)
PA3.update(AmbiguityPenalty=1, UnalignedPenalty=1)
HA = P(MutationPenalty=1, InsertionStart_Penalty=1.5, InsertionExtension_Penalty=0.6, DeletionStart_Penalty=1.5, DeletionExtension_Penalty=0.5,
       MaxErrorRate=0.1, AmbiguityPenalty=0.1, UnalignedPenalty=0.1, Max_PenaltySpan=0, MaxNumMatches=1)
HA9 = dict(HA)
HA9.update(MaxErrorRate=0.09)
HA5 = dict(HA)
HA5.update(MaxErrorRate=0.5)
local_cases = [
    dict(name="PathAligner/testQueryEndingWithMismatchAndExtension", cite="T/PathAligner_Test.java:10-15", chain=0, query="AACCGGTT", reference="AAT",
         params=PA, alignedA="AAC", alignedB="AAT", penalty=1.5, exact=True),
    dict(name="PathAligner/testQueryStartingWithShortExtension", cite="T/PathAligner_Test.java:17-26", chain=0, query="AAACCGGTTACGTACGTACGT",
         reference="AACCGGTTACGTTACGTACGT", params=PA, alignedA="AACCGGTTACG-TACGTACGT", alignedB="AACCGGTTACGTTACGTACGT", penalty=2.6, exact=True),
    dict(name="PathAligner/testMaxPenaltyHigherThanExtensionPenalty", cite="T/PathAligner_Test.java:28-39", chain=0,
         query="AACACACGGTGTTCACCACCCGCCCGCGCGCT", reference="AACCCACGGTGTTCACAATAACCGCCGGCGGT", params=PA3,
         alignedA="AACACACGGTGTTCACCACCCGCCCGCGCGCT", alignedB="AACCCACGGTGTTCACAATAACCGCCGGCGGT", penalty=10, exact=True),
    dict(name="HashBlockAligner/testQueryWithLongInsertion", cite="T/HashBlockAligner_Test.java:10-17", chain=1,
         query="GAGTGTCAATGACTGTTCGGCAACGGACATACTCCCGAACAGTCATTGACACTCCGTCCCACTCACGGAGAAGAGATTCTGCTGCAACCGGGCATCAACT",
         reference="AAAAAAAAACAGCGCAAAGAGCTGTTCGGCAACGGACATACTCCCGAATAGTCCTTGACACTCCGTCCCACTCACGGAGAAGAGATGCTGCTGCAACCGGGCATCAACTAAAAAAAAA",
         params=HA, alignedA="GAGTGTCAATGACTGTTCGGCAACGGACATACTCCCGAACAGTCATTGACACTCCGTCCCACTCACGGAGAAGAGATTCTGCTGCAACCGGGCATCAACT",
         alignedB="GAG---------CTGTTCGGCAACGGACATACTCCCGAATAGTCCTTGACACTCCGTCCCACTCACGGAGAAGAGATGCTGCTGCAACCGGGCATCAACT", penalty=9.9, exact=False),
    dict(name="HashBlockAligner/testInsertionCoveringThreeHashblocks", cite="T/HashBlockAligner_Test.java:19-26", chain=1,
         query="CACGCACAATGGCATGACAGCCAACAACAAAAGTAAAAAAATCGATTTTGTTCGCATGGTAGTATTAATAGGTTTATTGATGAAGCAAAGTGTGTCTCTTAAAGAAAT",
         reference="AAAAAAAAACACGCACAATGGCATGACAGCCAACAACAAAAGTAAAAAAATCGATTTTGTTCGCATGGTAGTATTAATAGGTTTATTGATGAAGCAAAGTAAAGAAATAAATCACTTTCCCGCCAAATTTAAAAAAAAA",
         params=HA, alignedA="CACGCACAATGGCATGACAGCCAACAACAAAAGTAAAAAAATCGATTTTGTTCGCATGGTAGTATTAATAGGTTTATTGATGAAGCAAAGTGTGTCTCTTAAAGAAAT",
         alignedB="CACGCACAATGGCATGACAGCCAACAACAAAAGTAAAAAAATCGATTTTGTTCGCATGGTAGTATTAATAGGTTTATTGATGAAGCAAAG---------TAAAGAAAT", penalty=6.9, exact=False),
    dict(name="HashBlockAligner/testQueryExtendingPastEndOfReference", cite="T/HashBlockAligner_Test.java:28-37", chain=1,
         query="TTTGATTCCTGTCTGATTCCCGTTCAATTCCCGCCAAGGTCCCACCGAGTTTTTTGCTTAAACCCCGTTTAATTTGCGTCAAGTTCCCGTTAAACTCCCT", reference="TTTGATTCCTGTCTGATTCCCG",
         params=HA9, alignedA="TTTGATTCCTGTCTGATTCCCG", alignedB="TTTGATTCCTGTCTGATTCCCG", penalty=7.8, exact=False),
    dict(name="HashBlockAligner/testQueryAlignedToMiddleOfReference", cite="T/HashBlockAligner_Test.java:39-48", chain=1, query="AACGT",
         reference="AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAACGTAAAAAAAAAAAAAA", params=HA5, alignedA="AACGT", alignedB="AACGT", penalty=0, exact=False),
]

symmetry_cases = [dict(cite="T/HashBlock_Test.java:12-28", text=t) for t in [
    "A", "C", "G", "T", "ACGTAACCGGTTACAGATCG",
    "TGTGTATATATAGCAAGAAGTGTCCTTGTCGGACAATTCTTGCTTTTCTCGCTTTGCTCAAAAAGATTTTAAGATTACCTTTGTGGCATGGAACTAAGACGGAACGAAAAGATTACATTCCGGTGTACCGAACTTGAAAAGGACGCACTT"]]

counting_cases = [
    dict(name="checkEfficientlyHandlesRepetitionInQuery", cite="T/Counting_HashBlockPath_Test.java:11-22", query="G" * 40,
         reference="GGGGGGGGACGTTGCAAACCGGTTATGCTGCAAATTGGCC", expectNum=0),
    dict(name="checkOneHashblockMatchSufficientNearEndOfReference", cite="T/Counting_HashBlockPath_Test.java:24-38", query="CCCTTAAGGACCGTGTGAGAACGAC",
         reference="ACGTAAGTACGAGCCGTAAGGTCCC", expectContainsOffset=12),
    dict(name="checkPoorAlignmentInsufficientEvenNearEndOfReference", cite="T/Counting_HashBlockPath_Test.java:40-54", query="GGACCCGG",
         reference="ACCCACCCACCCACCCACCC", expectNum=0),
]

paths_cases = [
    dict(name="checkComputesDistanceCorrectly", cite="T/HashBlockPaths_Counter_Test.java:12-18", reference="GGGGGACGTGGGGGGAACTAAGGGG", seq1="GACGTG",
         seq2="AACTAAG", num=1, inner=5, outer=18),
    dict(name="checkReverseComplementAlignment", cite="T/HashBlockPaths_Counter_Test.java:20-26", reference=rc("GGGGGACGTGGGGGGAACTAAGGGG"), seq1="GACGTG",
         seq2="AACTAAG", num=1, inner=5, outer=18),
    dict(name="checkOverlappingDistance", cite="T/HashBlockPaths_Counter_Test.java:28-34", reference="GGGGAACCACTGGGGG", seq1="GAACCACTG", seq2="CCACTGGGG",
         num=1, inner=-6, outer=12),
    dict(name="checkMultipleMatches", cite="T/HashBlockPaths_Counter_Test.java:36-46", reference="GGGGGAACAGTGGGGGGAACTAAGGGGAATTGTATATAGCG" * 2,
         seq1="GAACAGTG", seq2="AACTAAGGGGAA", num=2),
]

# T/SamWriter_Test.java:18-94 (DuplicationDetector(db, 1, 2, 2, 1); params = makeParameters() :113-124 == AW)
sam_cases = [
    dict(name="simpleTest", cite="T/SamWriter_Test.java:18-30", reference="ACGTAAAAACCGTAAA", mates=[["query", "ACGTA"]],
         sam="query\t0\tref\t1\t255\t5M\t*\t0\t5\tACGTA\t*\tAS:f:0.0\n"),
    dict(name="pairedEndAlignment", cite="T/SamWriter_Test.java:32-46", reference="AACCGGTTATAAAAAAAAAAACGTACGTATAAAAAAAAAA",
         mates=[["one", "AACCGGTTAT"], ["two", "ATACGTACGT"]], expectedInner=1, deviation=100,
         sam="one\t99\tref\t1\t255\t10M\tref\t21\t10\tAACCGGTTAT\t*\tcs:f:0.0\tAS:f:0.0\n"
             "two\t147\tref\t21\t255\t10M\tref\t1\t10\tACGTACGTAT\t*\tcs:f:0.0\tAS:f:0.0\n"),
    dict(name="oneReadWithMultipleAlignments", cite="T/SamWriter_Test.java:48-61", reference="ACGTAAAAACGTAAAA", mates=[["query", "ACGTA"]],
         sam="query\t0\tref\t1\t255\t5M\t*\t0\t5\tACGTA\t*\tAS:f:0.0\n"
             "query\t0\tref\t9\t255\t5M\t*\t0\t5\tACGTA\t*\tAS:f:0.0\n"),
    dict(name="pairedEndReadWithMultipleAlignments", cite="T/SamWriter_Test.java:63-79", reference="ACGTAAAACCCCCTTTTACGTAAAACCCCC",
         mates=[["one", "ACGTA"], ["two", "GGGGG"]], expectedInner=1, deviation=5,
         sam="one\t99\tref\t18\t255\t5M\tref\t26\t5\tACGTA\t*\tcs:f:0.0\tAS:f:0.0\n"
             "two\t147\tref\t26\t255\t5M\tref\t18\t5\tCCCCC\t*\tcs:f:0.0\tAS:f:0.0\n"
             "one\t99\tref\t1\t255\t5M\tref\t9\t5\tACGTA\t*\tcs:f:0.0\tAS:f:0.0\n"
             "two\t147\tref\t9\t255\t5M\tref\t1\t5\tCCCCC\t*\tcs:f:0.0\tAS:f:0.0\n"),
    dict(name="pairedEndAlignmentOnlyOneSequenceAligned", cite="T/SamWriter_Test.java:81-94", reference="AACCGGTTATAAAAAAAAAAACGTACGTATAAAAAAAAAA",
         mates=[["one", "AACCGGTTAT"], ["two", "CCCCCCCCCC"]], expectedInner=1, deviation=100,
         sam="one\t73\tref\t1\t255\t10M\t*\t0\t10\tAACCGGTTAT\t*\tcs:f:0.0\tAS:f:0.0\n"),
]

# T/BasepairsTest.java:9-45
basepairs_cases = [
    dict(cite="T/BasepairsTest.java:21-24", a="A", b="C", ambiguityPenalty=3, mutationPenalty=100, penalty=100.0),
    dict(cite="T/BasepairsTest.java:26-29", a="A", b="N", ambiguityPenalty=3, mutationPenalty=100, penalty=3.0),
    dict(cite="T/BasepairsTest.java:35-39", a="A", b="M", ambiguityPenalty=3, mutationPenalty=100, penalty=1.0),
    dict(cite="T/BasepairsTest.java:41-44", a="M", b="A", ambiguityPenalty=3, mutationPenalty=100, penalty=1.0),
]

# examples/ (config 1 plumbing): names state the expected outcome, no expected output file exists
examples = dict(
    cite="examples/reference.fasta, examples/queries.fasta, examples/test.sh:14",
    reference=[["contig1", "AAAACCAAAGGCTCGCGTA"], ["contig2", "ACGTAC"], ["contig3", "ACGTAACCGGTTAAACCCGGGTTTAAAACCCCGGGGTTTT"]],
    queries=[["query1-matches", "AAAACCAAAGG"], ["query2-1SNP", "AAAACCAAATG"], ["query3-matches", "ACGTAC"], ["query4-insertion", "AAAACCCAAAGG"],
             ["query5-deletion", "CCGGTTAAACCCGGTTTAAAACCCC"], ["query6-too-different", "ACGCGCTAAACCGAGG"]])

# T/MultiHashBlock_Test.java:12-77: an ambiguous text must offer, among its possibilities, the block of every text it can stand for
# (checkExpandingAmbiguities :79-82 replaces up to maxNumAmbiguities positions of `text` by every code that contains the base there;
# checkExpandingAmbiguitiesInto :84-133 is the pairwise check; the last pair is testHighlyAmbiguousShortSequence :67-77)
multi_cases = dict(
    cite="T/MultiHashBlock_Test.java:12-77",
    expanding=[dict(name="testShortAmbiguities", text="A", maxNumAmbiguities=1), dict(name="testMediumAmbiguities", text="AAA", maxNumAmbiguities=3),
               dict(name="testLongAmbiguity", text="AAAAAAAAAAAAAAA", maxNumAmbiguities=3), dict(name="testNonUniformAmbiguity", text="TTATGC", maxNumAmbiguities=1)],
    into=[[3 * b, b + code + b] for code, bases in (("R", "AG"), ("Y", "CT"), ("W", "AT"), ("S", "CG"), ("K", "GT"), ("M", "AC"), ("D", "AGT"), ("V", "ACG"), ("H", "ACT"), ("B", "CGT"))
          for b in bases] + [["AAAAAA", "ARRRRA"]])

# T/SequenceDatabase_Test.java:16-42: positions of many / of very long sequences encode and decode to themselves
codec_cases = dict(cite="T/SequenceDatabase_Test.java:16-42",
                   cases=[dict(name="testEncodingLargeSequences", numSequences=16, sequenceLength=1073741824),
                          dict(name="testEncodingManyLargeSequences", numSequences=8192, sequenceLength=2097152)])

# T/PackedMap_Test.java:13-49: buckets whose positions need more than 32 bits
packed_map_cases = dict(cite="T/PackedMap_Test.java:13-49",
                        note="8 repeating sequences of (int)Math.pow(2, 31) bases + reverse complements, PackedMap(5, 10, db, 1), blocks HashBlock(i, 1, i % 10, -(i % 10) - 1), "
                             "i < 20: get(i) = positions {i, i + 10}")

out = dict(align_cases=align_cases, local_cases=local_cases, symmetry_cases=symmetry_cases, counting_cases=counting_cases, paths_cases=paths_cases,
           sam_cases=sam_cases, basepairs_cases=basepairs_cases, examples=examples, multi_cases=multi_cases, codec_cases=codec_cases,
           packed_map_cases=packed_map_cases)
if __name__ == "__main__":
    import sys
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_reference.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(align_cases), "align cases")
