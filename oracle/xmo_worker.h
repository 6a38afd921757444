// ORACLE — test infrastructure only (see xmo_types.h).
// Restates the per-read driver: M/AlignerWorker.java:306-644 (alignToAncestralReference, getPenaltyLowerBound,
// quicklyConfidentInBestAlignment, getUnpairedAlignments) and the database assembly of M/Api.java:41-69 and
// M/Mapper.java:639-694.  The result cache (M/AlignmentCache.java) is output-neutral and not restated.
#pragma once
#include "xmo_extend.h"
#include "xmo_dup.h"

namespace xmo {

static inline double jmaxd(double a, double b) { if (a != a || b != b) return std::numeric_limits<double>::quiet_NaN(); return a > b ? a : b; }

// A query mate with its reverse complement (Sequence.reverseComplement() of a complement returns the original)
struct QuerySequence {
  std::unique_ptr<Sequence> fwd, rc;
  QuerySequence(const std::string& name, const std::vector<uint8_t>& codes) {
    fwd.reset(new Sequence());
    fwd->name = name;
    fwd->codes = codes;
    rc = makeReverseComplement(*fwd);
  }
};

struct ReferenceDatabase {  // M/ReferenceDatabase.java + M/Api.java:41-69 / M/Mapper.java:657-692
  SequenceDatabase sequences;
  std::unique_ptr<HashBlock_Database> hashblockDatabase;
  std::unique_ptr<DuplicationDetector> duplicationDetector;

  // apiMode=true:  Api.newDatabase (HashBlock_Database(refSequences), duplication window 1)
  // apiMode=false: Mapper.run      (hint max = chooseMaxDuplicationLength, duplication window 1000)
  // minInterestingSize > 0: the constructor argument of M/HashBlock_Database.java:34 (tests force the 13 a 3 Gb reference gets, :52, on a
  // reference the oracle can hash)
  void finish(bool apiMode, bool enableGapmers = true, int minInterestingSize = -1) {
    int minDup = HashBlock_Database::chooseMinDuplicationLength(sequences);
    int maxDup = HashBlock_Database::chooseMaxDuplicationLength(sequences);
    if (apiMode) hashblockDatabase.reset(new HashBlock_Database(&sequences, minInterestingSize, -1, -1, enableGapmers));
    else hashblockDatabase.reset(new HashBlock_Database(&sequences, minInterestingSize, maxDup, -1, enableGapmers));
    duplicationDetector.reset(new DuplicationDetector(hashblockDatabase.get(), minDup, maxDup, 2, apiMode ? 1 : 1000));
  }
  // SamWriter_Test-style assembly: DuplicationDetector(db, 1, 2, 2, 1)
  void finishCustomDup(int minDup, int maxDup, int copies, int window) {
    hashblockDatabase.reset(new HashBlock_Database(&sequences));
    duplicationDetector.reset(new DuplicationDetector(hashblockDatabase.get(), minDup, maxDup, copies, window));
  }
};

struct AlignerWorker {
  ReferenceDatabase* ref;
  AlignmentParameters parameters;
  Readable_HashBlock_Database referenceDatabase;
  const SequenceDatabase* sequenceDatabase;
  int shortestHashblockLength;
  Counters counters;

  AlignerWorker(ReferenceDatabase* ref, const AlignmentParameters& parameters)  // setup() :111-118
      : ref(ref), parameters(parameters), referenceDatabase(ref->hashblockDatabase.get()), sequenceDatabase(&ref->sequences),
        shortestHashblockLength(ref->hashblockDatabase->getMinInterestingSize()) {
    referenceDatabase.counters = &counters;
    ref->duplicationDetector->detect();  // duplicationDetector.helpSetup()
    referenceDatabase.prepare();         // process() :173
  }

  double getPenaltyLowerBound(int numMismatchedHashblocks) const {  // :487-491
    double mutationPenalty = numMismatchedHashblocks * parameters.MutationPenalty;
    double indelPenalty = shortestHashblockLength * numMismatchedHashblocks * parameters.DeletionExtension_Penalty;
    return std::min(mutationPenalty, indelPenalty);
  }

  bool quicklyConfidentInBestAlignment(const QueryAlignmentP& optimisticBestAlignment, const QueryMatch& optimisticBestMatch) {  // :494-587
    if (!optimisticBestAlignment) return false;
    if (optimisticBestAlignment->hasIndel()) return false;
    const Sequence* originalReference = optimisticBestMatch.getComponent(0).getSequenceB();
    int matchStart = optimisticBestMatch.getStartIndexB();
    int matchEnd = optimisticBestMatch.getEndIndexB();
    bool hasNearbyDuplication = false;
    double similarityDetectionGranularity = ref->duplicationDetector->getDetectionGranularity();
    double penalty = optimisticBestAlignment->getPenalty();
    double numberOfMutations = (penalty + parameters.Max_PenaltySpan) / parameters.MutationPenalty;
    double existingMutationRate = numberOfMutations / optimisticBestMatch.getQueryTotalLength();
    if (penalty <= 0 && parameters.Max_PenaltySpan < parameters.getMinPossibleNonzeroPenalty()) return true;
    double probabilityMutationInSection = 1 - std::pow(1 - existingMutationRate, similarityDetectionGranularity);
    double acceptableProbability = 1.0 / (double)sequenceDatabase->getTotalForwardAndReverseSize();
    double numberOfUnmatchedBlocksForHighConfidence = std::log(acceptableProbability) / std::log(probabilityMutationInSection);
    double totalLengthForHighConfidence = numberOfUnmatchedBlocksForHighConfidence * similarityDetectionGranularity;
    double matchMiddle = (double)((matchStart + matchEnd) / 2);
    double interestingWindow = jmaxd(totalLengthForHighConfidence, (double)((matchEnd - matchStart + 1) / 2));
    int windowStart = j2i(matchMiddle - interestingWindow);
    int windowEnd = j2i(matchMiddle + interestingWindow);
    if (ref->duplicationDetector->mayContainDuplicationInRange(originalReference, windowStart, windowEnd)) {
      hasNearbyDuplication = true;
    } else {
      if (matchStart <= interestingWindow) hasNearbyDuplication = true;
      else if (matchEnd >= originalReference->getLength() - interestingWindow) hasNearbyDuplication = true;
    }
    if (hasNearbyDuplication) return false;
    if (optimisticBestAlignment->hasAmbiguousBasepairs()) return false;
    return true;
  }

  // keeps the per-query objects alive while results are consumed
  struct QueryContext {
    std::vector<std::unique_ptr<QuerySequence>> mates;
    std::vector<std::unique_ptr<Counting_HashBlockPath>> components;
    std::vector<std::unique_ptr<QueryMatch_Aligner>> aligners;
    Query query;
  };

  // :306-484.  `ctx.mates` and `ctx.query` (sequences = mates[i]->fwd) must be filled by the caller.
  QueryAlignments alignToAncestralReference(QueryContext& ctx) {
    const Query& query = ctx.query;
    counters.reads++;
    double maxInterestingPenalty = query.getLength() * parameters.MaxErrorRate;
    int maxInnerDistance = j2i(maxInterestingPenalty * query.getSpacingDeviationPerUnitPenalty() + query.getExpectedInnerDistance());
    std::vector<Counting_HashBlockPath*> components;
    for (int i = 0; i < query.getNumSequences(); i++) {
      const Sequence* querySequence = ctx.mates[(size_t)i]->fwd.get();
      const Sequence* reverseComplementQuery = ctx.mates[(size_t)i]->rc.get();
      if (i > 0) std::swap(querySequence, reverseComplementQuery);  // :317-318
      ctx.components.emplace_back(new Counting_HashBlockPath(&referenceDatabase, sequenceDatabase, querySequence, reverseComplementQuery, parameters, &counters));
      components.push_back(ctx.components.back().get());
    }
    HashBlockPaths_Counter path(components, j2i(query.getExpectedInnerDistance()), maxInnerDistance);
    QueryAlignmentP optimisticBestAlignment;
    QueryMatchP optimisticBestMatch;
    int numMismatches = 0;
    QueryMatchListP bestMatches = path.optimisticGetBestMatches();
    ctx.aligners.emplace_back(new QueryMatch_Aligner(query, parameters, &counters));
    QueryMatch_Aligner& aligner = *ctx.aligners.back();
    QueryAlignments result;
    if (bestMatches->size() == 1) {
      optimisticBestMatch = (*bestMatches)[0];
      optimisticBestAlignment = aligner.align(*optimisticBestMatch, 0);
      if (quicklyConfidentInBestAlignment(optimisticBestAlignment, *optimisticBestMatch)) {
        counters.quickAccepts++;
        result.components.push_back(std::vector<QueryAlignmentP>(1, optimisticBestAlignment));  // QueryAlignments.singleChoice
        return result;
      }
    }
    if (optimisticBestAlignment) {
      while (true) {
        double possiblePenalty = getPenaltyLowerBound(numMismatches);
        if (possiblePenalty > optimisticBestAlignment->getPenalty() + parameters.Max_PenaltySpan) {
          result.components.push_back(std::vector<QueryAlignmentP>(1, optimisticBestAlignment));
          return result;
        }
        QueryMatchListP matches = path.findGoodPositionsHavingPriority(numMismatches);
        numMismatches++;
        bool done = false;
        for (auto& m : *matches) if (!optimisticBestMatch->samePosition(*m)) { done = true; break; }
        if (done) break;
      }
    }
    double bestPenalty = INT32_MAX;
    int candidateNumMismatches = 0;
    while (true) {
      double estimatedPenalty = getPenaltyLowerBound(candidateNumMismatches);
      if (estimatedPenalty > bestPenalty + parameters.Max_PenaltySpan) break;
      if (candidateNumMismatches > path.getNumBlocks()) break;
      QueryMatchListP candidates = path.findGoodPositionsHavingPriority(candidateNumMismatches);
      for (auto& match : *candidates) {
        QueryAlignmentP alignment;
        if (optimisticBestMatch && match->samePosition(*optimisticBestMatch)) alignment = optimisticBestAlignment;
        else alignment = aligner.align(*match, 0);
        if (alignment) {
          double penalty = alignment->getPenalty();
          if (bestPenalty > penalty) bestPenalty = penalty;
        }
      }
      if (estimatedPenalty >= maxInterestingPenalty) break;
      candidateNumMismatches++;
    }
    if (aligner.getBestAlignments().size() < 1 && query.getNumSequences() > 1) {
      QueryMatchListP partiallyGoodPositions = path.findPartiallyGoodPositions();
      for (auto& match : *partiallyGoodPositions) {
        QueryAlignmentP alignment = aligner.align(*match, 0);
        if (alignment) {
          double penalty = alignment->getPenalty();
          if (bestPenalty > penalty) bestPenalty = penalty;
        }
      }
    }
    std::vector<QueryAlignmentP> bestAlignments = aligner.getBestAlignments();
    result.components.push_back(bestAlignments);
    if (bestAlignments.size() < 1 && query.getNumSequences() > 1) result = getUnpairedAlignments(ctx, path);
    if ((int64_t)bestAlignments.size() > (int64_t)parameters.MaxNumMatches) {
      QueryAlignments unaligned;
      unaligned.components.push_back(std::vector<QueryAlignmentP>());
      return unaligned;
    }
    return result;
  }

  QueryAlignments getUnpairedAlignments(QueryContext& ctx, HashBlockPaths_Counter& path) {  // :602-644
    const Query& query = ctx.query;
    QueryAlignments out;
    out.components.resize(2);
    double expectedInnerDistance = query.getExpectedInnerDistance();
    for (int sequenceIndex = 0; sequenceIndex < query.getNumSequences(); sequenceIndex++) {
      const Sequence* sequence = query.getSequence(sequenceIndex);
      double maxInterestingSubqueryPenalty = sequence->getLength() * parameters.MaxErrorRate;
      int maxNumMutations = j2i(maxInterestingSubqueryPenalty / parameters.MutationPenalty);
      int maxNumMismatches = maxNumMutations;
      std::vector<SequenceMatchP> candidateLocations = path.findGoodComponentMatches(sequenceIndex, maxNumMismatches);
      Query subQuery = query.subquery(sequenceIndex);
      ctx.aligners.emplace_back(new QueryMatch_Aligner(subQuery, parameters, &counters));
      QueryMatch_Aligner& subqueryAligner = *ctx.aligners.back();
      for (auto& sequenceMatch : candidateLocations) {
        int minInnerDistance;
        if (sequenceIndex % 2 == 1) minInnerDistance = sequenceMatch->getStartIndexB();
        else minInnerDistance = sequenceMatch->sequenceB->getLength() - sequenceMatch->getEndIndexB();
        double innerDistance = minInnerDistance;
        if (innerDistance < expectedInnerDistance) innerDistance = expectedInnerDistance;
        double spacingPenalty = innerDistance / query.getSpacingDeviationPerUnitPenalty();
        if (spacingPenalty > maxInterestingSubqueryPenalty) continue;
        QueryMatch subqueryMatch(sequenceMatch, -1);
        subqueryAligner.align(subqueryMatch, innerDistance);
      }
      out.components[(size_t)sequenceIndex] = subqueryAligner.getBestAlignments();
    }
    return out;
  }
};

}  // namespace xmo
