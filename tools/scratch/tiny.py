import sys; sys.path.insert(0, '/root/repo')
import vsr_tlaplus_amd as vt
m1 = vt.Model.from_constants(R=2, C_=1, n=1, L=1)
for fw, fs in ((1 << 14, 1 << 10), (1 << 16, 1 << 12), (1 << 20, 1 << 14)):
    mc = vt.ModelChecker(m1, table_log2=12, frontier_words=fw, frontier_states=fs)
    try:
        d = mc.step()
        print(fw, fs, {k: d[k] for k in ("level", "n_new", "generated", "frontier", "words_new", "record_words")})
    except Exception as e:
        print(fw, fs, "ERR", e)
    mc.close()
