// SOURCE ONLY — not compiled or tested in this repository (no JDK / tla2tools.jar in the build image; see INTEGRATION.md).
// A tlc2.tool.fp.FPSet whose storage is the HBM-resident open-addressing table of libvsrmc.so (include/vsrmc.h).
// Select with -Dtlc2.tool.fp.FPSet.impl=tlc2.tool.fp.GpuFPSet.
package tlc2.tool.fp;

import java.io.IOException;
import java.rmi.RemoteException;
import java.util.ArrayDeque;
import java.util.concurrent.atomic.AtomicBoolean;
import java.util.concurrent.locks.Condition;
import java.util.concurrent.locks.ReentrantLock;

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

    // TLC's workers call put / contains one fingerprint at a time, from every worker thread.  One JNI call plus two PCIe copies per
    // fingerprint would cost more than TLC's own FPSet, so the calls are COMBINED: a caller files its request in a queue; whoever finds
    // no flush in progress becomes the leader, takes everything queued so far (up to BATCH requests of all threads), makes ONE native
    // call for the batch outside the lock, hands the answers back and wakes the others.  Two threads that put the same new fingerprint
    // in one batch get one "new" and one "present" (k_fpset_put claims a slot with one compare-and-swap), as with any FPSet.
    private static final class Req {
        final long fp;
        final boolean isPut;
        boolean done, present;
        IOException error;
        Req(long fp, boolean isPut) { this.fp = fp; this.isPut = isPut; }
    }

    private final ReentrantLock lock = new ReentrantLock();
    private final Condition flushed = lock.newCondition();
    private final ArrayDeque<Req> queue = new ArrayDeque<Req>();
    private boolean flushing = false;

    private boolean combined(long fp, boolean isPut) throws IOException {
        final Req r = new Req(fp, isPut);
        lock.lock();
        try {
            queue.add(r);
            while (!r.done) {
                if (flushing) {
                    flushed.awaitUninterruptibly();
                    continue;
                }
                flushing = true;                                  // this thread leads one flush: the oldest request decides the operation
                final boolean op = queue.peek().isPut;
                final ArrayDeque<Req> batch = new ArrayDeque<Req>();
                while (!queue.isEmpty() && batch.size() < BATCH && queue.peek().isPut == op) batch.add(queue.poll());
                lock.unlock();
                // Everything between this unlock and the re-lock below runs WITHOUT the lock, and whatever happens in it — a native
                // failure, a RuntimeException / Error out of the JNI call, an OutOfMemoryError allocating the arrays — the batch must be
                // marked done and `flushing` reset, or every other worker waits in awaitUninterruptibly() for ever.  The finally block
                // re-locks first (so that the outer finally's unlock() is balanced), then hands every request of the batch its answer or
                // the error.
                byte[] out = null;
                IOException err = null;
                try {
                    final int m = batch.size();
                    final long[] fps = new long[m];
                    out = new byte[m];
                    int i = 0;
                    for (Req q : batch) fps[i++] = q.fp;
                    final int rc = op ? putBlock0(handle, fps, out) : containsBlock0(handle, fps, out);
                    if (rc != 0) err = new IOException(lastError0());
                } catch (Throwable t) {
                    err = new IOException("GpuFPSet: the native batch call failed: " + t, t);
                } finally {
                    lock.lock();
                    int i = 0;
                    for (Req q : batch) {
                        q.present = err == null && out != null && out[i] != 0;
                        i++;
                        q.error = err;
                        q.done = true;
                    }
                    flushing = false;
                    flushed.signalAll();
                }
            }
        } finally {
            lock.unlock();
        }
        if (r.error != null) throw r.error;
        return r.present;
    }

    @Override
    public boolean put(long fp) throws IOException { return combined(fp, true); }

    @Override
    public boolean contains(long fp) throws IOException { return combined(fp, false); }

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
    private static final AtomicBoolean warned = new AtomicBoolean(false);
    @Override public void beginChkpt() {
        if (warned.compareAndSet(false, true))
            System.err.println("GpuFPSet: TLC checkpoints do NOT contain the fingerprint set (it lives in HBM and is not written out); "
                               + "a -recover of this run will be refused.  Use the native checker's -checkpoint / -recover instead.");
    }
    @Override public void commitChkpt() { }
    @Override public void recover() throws IOException {
        throw new IOException("GpuFPSet does not persist fingerprints across TLC checkpoints; resume with the native checker's -recover");
    }
    @Override public void beginChkpt(String filename) { beginChkpt(); }
    @Override public void commitChkpt(String filename) { commitChkpt(); }
    @Override public void recover(String filename) throws IOException { recover(); }
    @Override public void recoverFP(long fp) throws IOException { put(fp); }
}
