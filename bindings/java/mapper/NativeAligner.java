package mapper;

import java.nio.ByteBuffer;
import java.util.ArrayList;
import java.util.List;

/**
 * The per-read seed-and-extend path of X-Mapper on an AMD MI355X: a thin view of libxmapper_hip.so (include/xmapper_hip.h) through
 * libxmapper_jni.so (bindings/java/xmapper_jni.c).
 *
 * One instance per reference database: the index (HashBlock_Database tables, SequenceDatabase texts, DuplicationDetector keys) is built on
 * and lives in the GPU's memory.  align(List<Query>) replaces the per-query loop of AlignerWorker.process() (AlignerWorker.java:177-231):
 * it returns the same List<QueryAlignments>, built with the reference's own constructors, so AlignmentListeners (SamWriter, VcfWriter,
 * MutationsWriter, MatchDatabase, AlignmentCounter) see ordinary objects.
 *
 * This file is shipped as source for the maintainer (the development image of libxmapper_hip has no JDK); the C side it calls is the
 * tested boundary (tests/test_c_binding.py drives the same marshalling functions from plain C).
 */
final class NativeAligner implements AutoCloseable {
  static { System.loadLibrary("xmapper_jni"); }  // links against libxmapper_hip.so

  /** List<QueryAlignments> of one batch, flattened into two streams: the layout is documented at xm_result in include/xmapper_hip.h. */
  static final class ResultStreams {
    int[] ints;
    double[] dbls;
    long[] intOff;
    long[] dblOff;
    double kernelMillis;
  }

  private long indexHandle;                     // xm_index*
  private final List<Sequence> contigs;         // forward strands, in the order the index was built with
  private final AlignmentParameters parameters;

  /**
   * @param forwardContigs the reference contigs, forward strand only, in the order of Mapper.sortAndComplementReference (Mapper.java:1151-1172)
   * @param duplicationWindow 1000 for Mapper.run (Mapper.java:691), 1 for Api.newDatabase (Api.java:66)
   * @param maxQueryLength hint: the longest read expected (tables for longer reads are hashed on demand, as Readable_HashBlock_Database does)
   * @param device HIP device ordinal, -1 = current
   */
  NativeAligner(List<Sequence> forwardContigs, AlignmentParameters parameters, boolean enableGapmers, int duplicationWindow, int maxQueryLength, int device) {
    this.contigs = new ArrayList<Sequence>(forwardContigs);
    this.parameters = parameters;
    byte[][] codes = new byte[contigs.size()][];
    String[] names = new String[contigs.size()];
    for (int c = 0; c < contigs.size(); c++) {
      Sequence contig = contigs.get(c);
      byte[] text = new byte[contig.getLength()];
      for (int i = 0; i < text.length; i++) {
        text[i] = contig.encodedCharAt(i);        // 4-bit Basepairs code, one per byte
      }
      codes[c] = text;
      names[c] = contig.getName();
    }
    this.indexHandle = buildIndex(codes, names, enableGapmers, duplicationWindow, maxQueryLength, device);
  }

  private NativeAligner(NativeAligner other) {
    this.contigs = other.contigs;
    this.parameters = other.parameters;
    this.indexHandle = newContext(other.indexHandle);
  }

  /**
   * A view of the same index for another AlignerWorker thread, as HashBlock_Database.view() hands every worker its own
   * (HashBlock_Database.java:129-133): its own GPU stream, batch buffers and scratch; the tables are shared, nothing is copied.
   * Calls on one NativeAligner serialise, so a run with N workers on a GPU makes one NativeAligner and N - 1 views of it.
   */
  NativeAligner view() {
    return new NativeAligner(this);
  }

  /** The batch form of AlignerWorker.align(Query) (AlignerWorker.java:256-261): element q of the result belongs to batch.get(q). */
  List<QueryAlignments> align(List<Query> batch) {
    int nq = batch.size();
    int[] mateCount = new int[nq];
    long[] mateOffset = new long[2 * nq];
    int[] mateLength = new int[2 * nq];
    double[] expectedInner = new double[nq];
    double[] deviation = new double[nq];
    long total = 0;
    for (Query query : batch) {
      for (int m = 0; m < query.getNumSequences(); m++) {
        total += query.getSequence(m).getLength();
      }
    }
    if (total > Integer.MAX_VALUE - 16) {
      throw new IllegalArgumentException("batch of " + total + " bases: use batches below 2^31 bases");
    }
    ByteBuffer codes = ByteBuffer.allocateDirect((int)Math.max(total, 1));
    long offset = 0;
    for (int q = 0; q < nq; q++) {
      Query query = batch.get(q);
      if (query.getNumSequences() < 1 || query.getNumSequences() > 2) {
        throw new IllegalArgumentException("a query has 1 or 2 sequences: " + query.format());
      }
      mateCount[q] = query.getNumSequences();
      expectedInner[q] = query.getExpectedInnerDistance();
      deviation[q] = query.getSpacingDeviationPerUnitPenalty();
      for (int m = 0; m < mateCount[q]; m++) {
        Sequence mate = query.getSequence(m);     // as given in the FASTQ: mate 2 is NOT reverse-complemented here (AlignerWorker.java:317-318 does it)
        mateOffset[2 * q + m] = offset;
        mateLength[2 * q + m] = mate.getLength();
        for (int i = 0; i < mate.getLength(); i++) {
          codes.put((int)(offset + i), mate.encodedCharAt(i));
        }
        offset += mate.getLength();
      }
    }
    double[] p = new double[] {parameters.MutationPenalty, parameters.InsertionStart_Penalty, parameters.InsertionExtension_Penalty, parameters.DeletionStart_Penalty,
                               parameters.DeletionExtension_Penalty, parameters.MaxErrorRate, parameters.UnalignedPenalty, parameters.AmbiguityPenalty, parameters.Max_PenaltySpan};
    ResultStreams streams = new ResultStreams();
    if (!alignBatch(indexHandle, p, parameters.MaxNumMatches, mateCount, mateOffset, mateLength, codes, expectedInner, deviation, streams)) {
      throw new RuntimeException("Failed to align");  // (alignBatch has thrown already; this line is not reached)
    }
    List<QueryAlignments> results = new ArrayList<QueryAlignments>(nq);
    for (int q = 0; q < nq; q++) {
      results.add(decode(batch.get(q), streams, q));
    }
    return results;
  }

  /** Rebuilds the QueryAlignments of query q from the streams with the reference's own constructors. */
  private QueryAlignments decode(Query query, ResultStreams r, int q) {
    int i = (int)r.intOff[q];
    int d = (int)r.dblOff[q];
    int numComponents = r.ints[i++];
    List<List<QueryAlignment>> components = new ArrayList<List<QueryAlignment>>(numComponents);
    for (int c = 0; c < numComponents; c++) {
      int numAlignments = r.ints[i++];
      List<QueryAlignment> alignments = new ArrayList<QueryAlignment>(numAlignments);
      for (int a = 0; a < numAlignments; a++) {
        int innerDistance = r.ints[i++];
        int numSequences = r.ints[i++];
        double spacingPenalty = r.dbls[d++];
        double overlapMultiplier = r.dbls[d++];
        double duplicationBonus = r.dbls[d++];
        double totalPenalty = r.dbls[d++];
        List<SequenceAlignment> sequenceAlignments = new ArrayList<SequenceAlignment>(numSequences);
        for (int s = 0; s < numSequences; s++) {
          int contigIndex = r.ints[i++];
          boolean referenceReversed = r.ints[i++] != 0;
          int numBlocks = r.ints[i++];
          // which mate: a paired alignment lists mate 1 then mate 2; the unpaired fallback (AlignerWorker.java:602-644) has one component per mate
          int mateIndex = numComponents > 1 ? c : s;
          Sequence mate = query.getSequence(mateIndex);
          // the strand of the read that was aligned: SequenceMatch.getReversed() <=> its sequenceA is the reverse complement of the given one
          Sequence sequenceA = referenceReversed ? mate.reverseComplement() : mate;
          Sequence sequenceB = contigs.get(contigIndex);
          List<AlignedBlock> blocks = new ArrayList<AlignedBlock>(numBlocks);
          for (int b = 0; b < numBlocks; b++) {
            blocks.add(new AlignedBlock(sequenceA, sequenceB, r.ints[i], r.ints[i + 1], r.ints[i + 2], r.ints[i + 3]));
            i += 4;
          }
          // the constructor AlignmentParameters.newSequenceAlignment ends with (AlignmentParameters.java:94), given the penalties the GPU path
          // computed with the reference's order of additions (total = aligned + unaligned bases x UnalignedPenalty)
          double totalSequencePenalty = r.dbls[d++];
          double alignedPenalty = r.dbls[d++];
          SequenceAlignment alignment = new SequenceAlignment(blocks, referenceReversed, totalSequencePenalty, alignedPenalty);
          sequenceAlignments.add(alignment);
        }
        alignments.add(new QueryAlignment(sequenceAlignments, spacingPenalty, overlapMultiplier, duplicationBonus, totalPenalty, innerDistance));  // QueryMatch_Aligner.java:267
      }
      components.add(alignments);
    }
    if (numComponents == 1) {
      if (components.get(0).size() < 1) {
        return QueryAlignments.unaligned(query.getSequences());                     // AlignerWorker.java:480
      }
      return QueryAlignments.singleComponent(query.getSequences(), components.get(0));  // AlignerWorker.java:471
    }
    return new QueryAlignments(query.getSequences(), components);                    // AlignerWorker.java:643
  }

  @Override public void close() {
    if (indexHandle != 0) {
      freeIndex(indexHandle);
      indexHandle = 0;
    }
  }

  private static native long buildIndex(byte[][] contigCodes, String[] names, boolean enableGapmers, int duplicationWindow, int maxQueryLength, int device);
  private static native long newContext(long handle);
  private static native void freeIndex(long handle);
  private static native boolean alignBatch(long handle, double[] parameters9, int maxNumMatches, int[] mateCount, long[] mateOffset, int[] mateLength, ByteBuffer codes,
                                           double[] expectedInnerDistance, double[] spacingDeviationPerUnitPenalty, ResultStreams out);
}
