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
    // ---- one engine over ALL the GPUs of a node (kvm_group, include/kvmatch.h): the series on every device, the candidate list
    // of a call dealt out to them, ONE RCCL all-gather of the answers issued by the library -- one call per query, the result
    // the single-GPU call returns.  `devices`: HIP ordinals of this process; inFlight: queries a server's threads keep going.
    public static native long groupOpen(int[] devices, int inFlight) throws IOException;
    /** One JVM per GPU instead: rank `rank` of `world`; `id` = groupUniqueId() of ONE rank, carried to the others by the host. */
    public static native long groupOpenRank(int device, int rank, int world, byte[] id, int inFlight) throws IOException;
    public static native byte[] groupUniqueId() throws IOException;
    public static native void groupLoadSeriesFile(long group, String path, long n) throws IOException;
    /** variant 0 rsmEd, 1 cnsmEd, 2 rsmDtw, 3 cnsmDtw; rho / alpha / beta ignored where the variant has none. */
    public static native Result groupVerify(long group, int variant, double[] q, int[] left, int[] right, int shift, double eps, int rho, double alpha, double beta);
    public static native void groupSetOption(long group, int option, long value);
    public static native void groupClose(long group);
    /** The engine-side handle: `new Phase2Gpu.Group(new int[] {0,1,2,3,4,5,6,7}, dataFile, n)` where an engine holds a Phase2Gpu. */
    public static final class Group implements AutoCloseable {
        private long g;
        public Group(int[] devices, String dataFile, int n) throws IOException { g = groupOpen(devices, 1); groupLoadSeriesFile(g, dataFile, n); }
        public Result rsmEd  (double[] q, int[] left, int[] right, int shift, double eps) { return groupVerify(g, 0, q, left, right, shift, eps, 0, 0, 0); }
        public Result cnsmEd (double[] q, int[] left, int[] right, int shift, double eps, double alpha, double beta) { return groupVerify(g, 1, q, left, right, shift, eps, 0, alpha, beta); }
        public Result rsmDtw (double[] q, int[] left, int[] right, int shift, double eps, int rho) { return groupVerify(g, 2, q, left, right, shift, eps, rho, 0, 0); }
        public Result cnsmDtw(double[] q, int[] left, int[] right, int shift, double eps, int rho, double alpha, double beta) { return groupVerify(g, 3, q, left, right, shift, eps, rho, alpha, beta); }
        @Override public void close() { if (g != 0) { groupClose(g); g = 0; } }
    }
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
