// Phase2LoopsBaseline.java -- TEST / MEASUREMENT INFRASTRUCTURE (lives under oracle/): a minimal Java transcription of the
// RSM-ED phase-2 loop (the block between `startTime2` and `endTime2` of cn.edu.fudan.dsm.kvmatch.QueryEngine.query,
// K/QueryEngine.java:338-363) over boxed List<Double>, the way the reference holds its data -- for a box that has a JDK, so
// that bench.py can report a CPU number that pays for Java's boxing like the reference does (BASELINE.md section 3, item 2;
// the C port of oracle/kvm_oracle.c avoids boxing and therefore flatters the CPU side).  It is NOT the reference: the
// reference cannot be built without its Maven dependencies; this file restates the one loop, single-threaded like it.
// Never compiled in the build image (no JDK there): bench.py compiles and runs it only where `javac` and `java` exist.
//
//   javac -d <dir> Phase2LoopsBaseline.java
//   java -cp <dir> Phase2LoopsBaseline <data file: big-endian doubles> <n> <intervals file: little-endian int32 pairs> \
//        <query offset, 1-based> <m> <epsilon> <earlyAbandon: 0|1>
//   prints one line:  candidates <cntCandidate> windows <evaluated> answers <count> seconds <t>
import java.io.IOException;
import java.io.RandomAccessFile;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.file.Files;
import java.nio.file.Paths;
import java.util.ArrayList;
import java.util.List;

public class Phase2LoopsBaseline {
    private static RandomAccessFile dataFile;
    private static int n;

    // readTimeSeries(left, length) of the file operator: `length` values from 1-based offset `left`, boxed one by one
    // (K/operator/file/TimeSeriesFileOperator.java:54-96 decodes big-endian doubles into a List<Double>)
    private static List<Double> readTimeSeries(int left, int length) throws IOException {
        if (left < 1 || left + length - 1 > n || length < 1)
            throw new IllegalArgumentException("left: " + left + ", length: " + length);
        byte[] raw = new byte[length * 8];
        dataFile.seek((long) (left - 1) * 8L);
        dataFile.readFully(raw);
        ByteBuffer bb = ByteBuffer.wrap(raw).order(ByteOrder.BIG_ENDIAN);
        List<Double> out = new ArrayList<>(length);
        for (int i = 0; i < length; i++) out.add(bb.getDouble());
        return out;
    }

    public static void main(String[] args) throws IOException {
        if (args.length < 7) {
            System.err.println("usage: Phase2LoopsBaseline <data> <n> <intervals> <queryOffset> <m> <epsilon> <earlyAbandon>");
            System.exit(2);
        }
        n = Integer.parseInt(args[1]);
        dataFile = new RandomAccessFile(args[0], "r");
        byte[] iv = Files.readAllBytes(Paths.get(args[2]));
        ByteBuffer ib = ByteBuffer.wrap(iv).order(ByteOrder.LITTLE_ENDIAN);
        int k = iv.length / 8;
        int[] left = new int[k], right = new int[k];
        for (int i = 0; i < k; i++) {
            left[i] = ib.getInt();
            right[i] = ib.getInt();
        }
        int queryOffset = Integer.parseInt(args[3]), m = Integer.parseInt(args[4]);
        double epsilon = Double.parseDouble(args[5]);
        boolean earlyAbandon = Integer.parseInt(args[6]) != 0;
        List<Double> queryData = readTimeSeries(queryOffset, m);

        long t0 = System.nanoTime();
        long cntCandidate = 0, windows = 0;
        List<double[]> answers = new ArrayList<>();
        final int shift = 0;                                   // lastSegment = 1: (lastSegment - 1) * WuList[0]
        for (int p = 0; p < k; p++) {
            cntCandidate += right[p] - left[p] + 1;            // before clipping
            int begin = left[p] - shift;
            int end = right[p] - shift + m - 1;
            if (begin < 1) begin = 1;
            if (end > n) end = n;
            List<Double> data = readTimeSeries(begin, end - begin + 1);
            for (int i = 0; i + m - 1 < data.size(); i++) {
                double dist = 0;
                for (int j = 0; j < m && (!earlyAbandon || dist <= epsilon * epsilon); j++)
                    dist += (data.get(i + j) - queryData.get(j)) * (data.get(i + j) - queryData.get(j));
                windows++;
                if (dist <= epsilon * epsilon) answers.add(new double[] {begin + i, Math.sqrt(dist)});
            }
        }
        double seconds = (System.nanoTime() - t0) * 1e-9;
        System.out.println("candidates " + cntCandidate + " windows " + windows + " answers " + answers.size() + " seconds " + seconds);
        dataFile.close();
    }
}
