/* hevcasm self-test executable (reference src/bin/main.c:39-41). */
#include "hevcasm.h"

int main(int argc, const char *argv[])
{
	return hevcasm_main(argc, argv);
}
