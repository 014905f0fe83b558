// SOURCE ONLY — not compiled or tested in this repository (no JDK in the build image; see INTEGRATION.md §2).
// A main() that keeps TLC's command line for this model (`-config VSR.cfg VSR.tla [-deadlock] [-simulate] [-depth N]`) and runs the
// whole Worker / StateQueue / FPSet / TLCTrace loop on the GPU through the C ABI of include/vsrmc.h (JNI glue: the natives below
// map 1:1 onto vsrmc_model_load, vsrmc_checker_create, vsrmc_checker_step, vsrmc_checker_trace, vsrmc_model_format_state,
// vsrmc_simulate, vsrmc_model_replay).
package tlc2.tool;

public final class GpuModelChecker {
    static { System.loadLibrary("vsrmc_jni"); }

    // --- natives (java/jni/vsrmc_jni.c) --------------------------------------------------------------------------------
    private static native long modelLoad(String tlaPath, String cfgPath);                 // vsrmc_model_load
    private static native long checkerCreate(long model, int device, int tableLog2, long frontierWords,
                                             long frontierStates, long pendingEntries, long traceEntries);   // vsrmc_checker_create
    /** one BFS level; out = {level, nNew, distinct, totalGenerated, violMask, violIndex, deadlocks}; returns the C ABI code */
    private static native int checkerStep(long checker, long[] out);                      // vsrmc_checker_step
    private static native String[] checkerTrace(long checker, long model, int level, long index);   // vsrmc_checker_trace + format_state
    private static native String[] simulate(long model, int device, int walkers, int depth, long seed, double maxSeconds);
    private static native String lastError();                                             // vsrmc_last_error

    public static void main(String[] args) {
        String cfg = null, tla = null;
        boolean simulate = false, checkDeadlock = false;
        int depth = 100;
        for (int i = 0; i < args.length; i++) {
            switch (args[i]) {
                case "-config": cfg = args[++i]; break;
                case "-deadlock": checkDeadlock = false; break;      // TLC: "-deadlock" = do NOT check for deadlock
                case "-simulate": simulate = true; break;
                case "-depth": depth = Integer.parseInt(args[++i]); break;
                case "-workers": i++; break;                          // the GPU is the worker pool
                default: if (!args[i].startsWith("-")) tla = args[i];
            }
        }
        long model = modelLoad(tla, cfg);
        if (model == 0) { System.err.println("Error: " + lastError()); System.exit(1); }
        if (simulate) {
            String[] trace = simulate(model, 0, 1 << 17, depth, 1L, 600.0);
            if (trace != null) {
                System.out.println("Error: Invariant AcknowledgedWriteNotLost is violated.\nError: The behavior up to this point is:");
                for (String s : trace) System.out.println(s);
                System.exit(12);
            }
            System.exit(0);
        }
        long mc = checkerCreate(model, 0, 30, 1L << 32, 1L << 27, 1L << 28, 1L << 29);
        if (mc == 0) { System.err.println("Error: " + lastError()); System.exit(1); }
        long[] info = new long[7];
        System.out.println("Finished computing initial states: 1 distinct state generated.");
        while (true) {
            int rc = checkerStep(mc, info);
            if (rc != 0) { System.out.println("Error: " + lastError()); System.exit(rc == -4 ? 12 : 1); }
            System.out.printf("Progress(%d): %d states generated, %d distinct states found, %d states left on queue.%n",
                              info[0], info[3], info[2], info[1]);
            if (info[4] != 0) {
                System.out.println("Error: Invariant AcknowledgedWriteNotLost is violated.\nError: The behavior up to this point is:");
                for (String s : checkerTrace(mc, model, (int) info[0], info[5])) System.out.println(s);
                System.exit(12);
            }
            if (checkDeadlock && info[6] != 0) { System.out.println("Error: Deadlock reached."); System.exit(11); }
            if (info[1] == 0) { System.out.println("Model checking completed. No error has been found."); break; }
        }
    }
}
