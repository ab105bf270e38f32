/*
 * Phase2Gpu.java -- the Java side of the drop-in (INTEGRATION.md section 2): thin natives over libkvmatch_jni.so
 * (kv-match_amd/jni/kvmatch_jni.c), which wraps the C ABI of include/kvmatch.h.  Shipped as source: this repository's build
 * image has no JDK; a maintainer drops the file into src/main/java of KV-match and compiles it with the project.
 */
package cn.edu.fudan.dsm.kvmatch.gpu;

import cn.edu.fudan.dsm.kvmatch.common.Pair;
import java.io.IOException;
import java.util.ArrayList;
import java.util.List;

/** One instance per engine (engines are not re-entrant, neither is a kvm_ctx). */
public final class Phase2Gpu implements AutoCloseable {
    static { System.loadLibrary("kvmatch_jni"); }
    private long ctx;                                   // kvm_ctx*

    public Phase2Gpu(int device, String dataFile, int n) throws IOException {
        ctx = open(device);
        loadSeriesFile(ctx, dataFile, n);               // files/data-{N}, staged once (kvm_load_series_file)
    }
    // returns offsets[] / distances[] already sorted by distance; stats = {cntCandidate}
    public static final class Result { public int[] offsets; public double[] distances; public long cntCandidate; }

    public native Result rsmEd  (long ctx, double[] q, int[] left, int[] right, int shift, double eps);
    public native Result cnsmEd (long ctx, double[] q, int[] left, int[] right, int shift, double eps, double alpha, double beta);
    public native Result rsmDtw (long ctx, double[] q, int[] left, int[] right, int shift, double eps, int rho);
    public native Result cnsmDtw(long ctx, double[] q, int[] left, int[] right, int shift, double eps, int rho, double alpha, double beta);
    // optional: TimeSeriesOperator.readTimeSeries from HBM; index build step 1 (see the end of section 3)
    public static native double[] readTimeSeries(long ctx, int left, int length);
    public static native double phase2TimeEstimateMs(int variant, long disjointWindows, long candidateOffsets, int length);
    public static native boolean phase1ShouldStop(int variant, int segmentIndex, double phase1Ms, long disjointWindows, long candidateOffsets, int length, double[] last);
    public static native long indexRowsBuild(long ctx, int w);
    public static native long indexRowsBuildGrouped(long ctx, int w);
    public static native long indexRowsGroups(long rows);
    public static native void indexRowsGetGroups(long rows, double[] keys, int[] offsets);
    public static native long indexRowsCount(long rows);
    public static native void indexRowsGet(long rows, long from, double[] keys, int[] first, int[] last);
    public static native void indexRowsFree(long rows);
    /** The bytes IndexBuilder.SingleIndexBuilder.run + IndexFileOperator.writeAll produce for window length w. */
    public static native byte[] indexBuild(long ctx, int w);
    /** kvm_set_option; option 6 (KVM_OPT_SORT_VIOLATION) = 1: answer cNSM-DTW queries on which Arrays.sort throws;
     *  option 7 (KVM_OPT_UCR_LOOPS) = 1 around a full scan: the tie behaviour of Ucr{Ed,Dtw}QueryExecutor instead of the engines';
     *  option 8 (KVM_OPT_STRICT_BOUNDS, default 1) = 0: answer where DtwUtils.lowerUpperLemire ends in ArrayIndexOutOfBounds;
     *  option 9 (KVM_OPT_IN_FLIGHT, default 1) = how many contexts the server's threads keep busy on this GPU at once: a hint to
     *  the ED kernel's launcher (throughput instead of the shortest single launch), the answers are the same. */
    public static native void setOption(long ctx, int option, long value);
    private static native long open(int device) throws IOException;
    private static native void loadSeriesFile(long ctx, String path, long n) throws IOException;
    private static native void close(long ctx);
    public long handle() { return ctx; }
    @Override public void close() { if (ctx != 0) { close(ctx); ctx = 0; } }

    public static List<Pair<Integer, Double>> toAnswers(Result r) {
        List<Pair<Integer, Double>> a = new ArrayList<>(r.offsets.length);
        for (int i = 0; i < r.offsets.length; i++) a.add(new Pair<>(r.offsets[i], r.distances[i]));
        return a;
    }
}
