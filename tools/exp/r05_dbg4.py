import faulthandler, sys, pathlib, tempfile
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
faulthandler.dump_traceback_later(12, repeat=True)
import train
d = tempfile.mkdtemp()
train.main(["SYN", "--synthetic", "8", "-b", "4", "--epochs", "2", "--img-height", "64", "--img-width", "96", "--lr", "1e-3", "--save-root", d,
            "--print-freq", "100", "--network", "disp_vgg_BN", "--with-gt"] + sys.argv[1:])
