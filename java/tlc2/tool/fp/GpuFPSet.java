// SOURCE ONLY — not compiled or tested in this repository (no JDK / tla2tools.jar in the build image; see INTEGRATION.md).
// A tlc2.tool.fp.FPSet whose storage is the HBM-resident open-addressing table of libvsrmc.so (include/vsrmc.h).
// Select with -Dtlc2.tool.fp.FPSet.impl=tlc2.tool.fp.GpuFPSet.
package tlc2.tool.fp;

import java.io.IOException;
import java.rmi.RemoteException;

import tlc2.util.BitVector;
import tlc2.util.LongVec;

@SuppressWarnings("serial")
public class GpuFPSet extends FPSet {
    static { System.loadLibrary("vsrmc_jni"); }

    private static final int BATCH = 1 << 16;
    private long handle;                       // vsrmc_fpset*
    private final int log2Slots;

    private static native long create0(int device, int log2Slots);
    private static native int putBlock0(long handle, long[] fps, byte[] wasPresent);
    private static native int containsBlock0(long handle, long[] fps, byte[] present);
    private static native long size0(long handle);
    private static native void destroy0(long handle);
    private static native String lastError0();

    public GpuFPSet(final FPSetConfiguration cfg) throws RemoteException {
        super(cfg);
        // 16-byte slots, load factor <= 0.5: slots = 2^ceil(log2(memory / 16))
        long slots = Math.max(1L << 20, cfg.getMemoryInBytes() / 16);
        this.log2Slots = 63 - Long.numberOfLeadingZeros(slots);
    }

    @Override
    public FPSet init(int numThreads, String metadir, String filename) throws IOException {
        handle = create0(Integer.getInteger("vsrmc.device", 0), log2Slots);
        if (handle == 0) throw new IOException(lastError0());
        return this;
    }

    // TLC's workers call put/contains one fingerprint at a time; each call is a batch of one under the monitor.
    // (The intended use is putBlock/containsBlock, or replacing the whole loop — INTEGRATION.md §2.)
    @Override
    public synchronized boolean put(long fp) throws IOException {
        byte[] out = new byte[1];
        if (putBlock0(handle, new long[] {fp}, out) != 0) throw new IOException(lastError0());
        return out[0] != 0;
    }

    @Override
    public synchronized boolean contains(long fp) throws IOException {
        byte[] out = new byte[1];
        if (containsBlock0(handle, new long[] {fp}, out) != 0) throw new IOException(lastError0());
        return out[0] != 0;
    }

    @Override
    public synchronized BitVector putBlock(LongVec fpv) throws IOException {
        return block(fpv, true);
    }

    @Override
    public synchronized BitVector containsBlock(LongVec fpv) throws IOException {
        return block(fpv, false);
    }

    private BitVector block(LongVec fpv, boolean put) throws IOException {
        int n = fpv.size();
        BitVector bv = new BitVector(n);
        for (int base = 0; base < n; base += BATCH) {
            int m = Math.min(BATCH, n - base);
            long[] fps = new long[m];
            for (int i = 0; i < m; i++) fps[i] = fpv.elementAt(base + i);
            byte[] out = new byte[m];
            int rc = put ? putBlock0(handle, fps, out) : containsBlock0(handle, fps, out);
            if (rc != 0) throw new IOException(lastError0());
            for (int i = 0; i < m; i++) if (out[i] == 0) bv.set(base + i);   // TLC convention: bit set = NOT present
        }
        return bv;
    }

    // NOTE: fingerprint 0 is the device table's empty-slot sentinel; put / contains remap it to 1 (k_fpset_put, k_fpset_contains), so
    // the two TLC fingerprints 0 and 1 share a slot — one possible false "seen" in 2^64, the same order as a fingerprint collision.
    @Override
    public long size() { return size0(handle); }

    @Override
    public long checkFPs() { return Long.MAX_VALUE; }   // collision-distance estimate: not computed on the device

    @Override
    public void close() { if (handle != 0) { destroy0(handle); handle = 0; } }

    @Override
    public void exit(boolean cleanup) throws IOException { close(); }

    // TLC checkpoints periodically by default (every 30 minutes): these hooks must not abort a long run.  The set lives in HBM
    // and this shim does not persist it — begin / commit are no-ops (a `-recover` of such a checkpoint finds an empty set and is
    // refused below); the native checker's own checkpoint (vsrmc_checker_save / _load) is the supported way to stop and resume.
    @Override public void beginChkpt() { }
    @Override public void commitChkpt() { }
    @Override public void recover() throws IOException {
        throw new IOException("GpuFPSet does not persist fingerprints across TLC checkpoints; resume with the native checker's -recover");
    }
    @Override public void beginChkpt(String filename) { beginChkpt(); }
    @Override public void commitChkpt(String filename) { commitChkpt(); }
    @Override public void recover(String filename) { recover(); }
    @Override public void recoverFP(long fp) throws IOException { put(fp); }
}
